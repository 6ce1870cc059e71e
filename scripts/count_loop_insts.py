#!/usr/bin/env python3
"""Static instruction mix of a kernel's hottest loop, from the -save-temps assembly (`make -C snowmocap_amd/csrc asm`).

    python scripts/count_loop_insts.py <kernel-name-substring> [rcp+rsq per item, e.g. 8 for 4 cameras]

Picks the innermost loop that contains v_rsq_f64 / v_rcp_f64 (the item loop of k_fused_single, the candidate loops of
k_frame_recompute) and prints VALU / fp64 / transcendental / LDS / VMEM / SALU counts per loop body and per item.
"""
import re
import sys

import os
ASM = os.environ.get("SNOWTRI_ASM", "snowmocap_amd/csrc/build/snowtri-hip-amdgcn-amd-amdhsa-gfx950.s")


def kernel_lines(name):
    out, on = [], False
    for ln in open(ASM):
        if not on and re.match(r"^_Z\w*:", ln) and name in ln:
            on = True
        if on:
            out.append(ln.rstrip("\n"))
            if ln.startswith(".Lfunc_end"):
                break
    return out


def classify(ins):
    op = ins.split()[0]
    if op.startswith("v_"):
        kind = "valu"
        if "_f64" in op:
            kind = "valu_f64"
        if re.match(r"v_(rcp|rsq|sqrt|exp|log|sin|cos)_", op):
            kind = "valu_trans"
        if op.startswith(("v_readlane", "v_writelane", "v_readfirstlane")):
            kind = "valu_lane"
        return kind
    if op.startswith("ds_"):
        return "lds"
    if op.startswith(("global_", "buffer_", "flat_", "scratch_")):
        return "vmem"
    if op.startswith("s_waitcnt"):
        return "waitcnt"
    if op.startswith("s_"):
        return "salu"
    return "other"


def main():
    name = sys.argv[1]
    per_item = int(sys.argv[2]) if len(sys.argv) > 2 else 0
    lines = kernel_lines(name)
    if not lines:
        sys.exit("kernel not found")
    # loops: label ... backward branch to that label
    labels = {m.group(1): i for i, ln in enumerate(lines) if (m := re.match(r"^(\.LBB\d+_\d+):", ln))}
    loops = []
    for i, ln in enumerate(lines):
        m = re.search(r"s_cbranch_\w+\s+(\.LBB\d+_\d+)", ln) or re.search(r"s_branch\s+(\.LBB\d+_\d+)", ln)
        if m and m.group(1) in labels and labels[m.group(1)] < i:
            loops.append((labels[m.group(1)], i))
    def n_trans(a, b):
        return sum(1 for ln in lines[a:b + 1] if re.search(r"\bv_(rsq|rcp)_f64", ln))
    most = max((n_trans(a, b) for a, b in loops), default=0)
    if not most:
        sys.exit("no loop with v_rsq_f64 / v_rcp_f64")
    # the tightest loop that still holds at least half of the reciprocals / rsqrts of the biggest one
    best = min(((a, b) for a, b in loops if 2 * n_trans(a, b) >= most), key=lambda ab: ab[1] - ab[0])
    a, b = best[:2]
    counts = {}
    for ln in lines[a:b + 1]:
        t = ln.strip()
        if not t or t.startswith((";", ".")) or t.endswith(":"):
            continue
        k = classify(t)
        counts[k] = counts.get(k, 0) + 1
    valu = sum(v for k, v in counts.items() if k.startswith("valu"))
    unroll = max(1, round((counts.get("valu_trans", 0)) / per_item)) if per_item else 1
    print(f"{name}: loop lines {a}-{b} of the kernel, {unroll} item(s) per loop body")
    for k in sorted(counts):
        print(f"  {k:11s} {counts[k]:5d}   per item {counts[k] / unroll:7.1f}")
    print(f"  VALU total  {valu:5d}   per item {valu / unroll:7.1f}")


if __name__ == "__main__":
    main()
