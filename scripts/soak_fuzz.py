#!/usr/bin/env python3
"""Dev aid (GPU box): soak run of the randomised parity sweeps of tests/test_gpu_parity.py with many seeds.
usage: python scripts/soak_fuzz.py <rounds> [first_round] [--only=<part of a sweep's name>]   -- every round re-seeds each sweep (round r: seed * 100003 + r + 1);
the first failures are printed.  first_round continues an earlier soak with fresh seeds."""
import os, sys, time, traceback
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import snowmocap_amd as api
import test_gpu_parity as T
import test_gpu_lean as TL
import test_gpu_handover as TH
import test_gpu_single_rigs as TS

real_rng = np.random.default_rng


import conftest


class MP:                        # what conftest.Knobs needs of pytest's monkeypatch
    def setenv(self, k, v): os.environ[k] = v
    def delenv(self, k, raising=True): os.environ.pop(k, None)


def Env():                       # the `knobs` fixture of the tests: a forced route binds the test build of the library (conftest.Knobs)
    k = conftest.Knobs(MP())
    k.undo = lambda: (k.clear(), k.restore())
    return k


_pos = [a for a in sys.argv[1:] if not a.startswith("--")]
rounds = int(_pos[0]) if len(_pos) > 0 else 20
first = int(_pos[1]) if len(_pos) > 1 else 0
sweeps = [("small_rigs auto", lambda e: T.test_random_small_rigs_against_oracle(api, "auto", e)),
          ("small_rigs spill", lambda e: T.test_random_small_rigs_against_oracle(api, "spill", e)),
          ("special auto", lambda e: T.test_random_special_values_against_oracle(api, "auto", e)),
          ("special spill", lambda e: T.test_random_special_values_against_oracle(api, "spill", e)),
          ("single person", lambda e: T.test_random_single_person_fast_path_and_fallback(api)),
          ("dlt multi", lambda e: T.test_random_small_rigs_dlt_against_oracle(api)),
          ("per-frame api", lambda e: T.test_random_small_rigs_per_frame_api(api)),
          ("lean thresholds", lambda e: TL.test_lean_random_thresholds_and_person_lists(api)),
          ("lean special", lambda e: TL.test_lean_special_values(api)),
          ("handover", lambda e: TH.test_random_rigs_with_handover_against_oracle_and_phase3(api, e)),
          ("handover wide rigs", lambda e: TH.test_random_wide_rigs_against_oracle_and_phase3(api, e)),
          ("float64 outputs / keypoint_num on the streaming route", lambda e: TH.test_random_rigs_float64_outputs_and_keypoint_num(api, e)),
          ("single-person rigs of 5-16 cameras, every route", lambda e: TS.test_random_single_person_rigs_on_every_route(api))]
only = [a.split("=", 1)[1] for a in sys.argv if a.startswith("--only=")]
if only:
    sweeps = [sw for sw in sweeps if any(o in sw[0] for o in only)]
fails = 0
t0 = time.time()
for r in range(first, first + rounds):
    np.random.default_rng = lambda seed=None, r=r: real_rng(None if seed is None else int(seed) * 100003 + r + 1)
    for name, fn in sweeps:
        env = Env()
        try:
            fn(env)
        except AssertionError as ex:
            msg = str(ex)
            if msg.strip() == "" or msg.strip().startswith(("(", "[")):     # bare coverage asserts of the fixed-seed tests (route counters)
                pass                                    # coverage counters of the fixed-seed test, not a parity failure
            else:
                fails += 1
                print(f"FAIL round {r} sweep {name}: {msg[:900]}")
                if fails >= 12:
                    print("stopping after 12 failures"); sys.exit(1)
        except Exception:
            fails += 1
            print(f"ERROR round {r} sweep {name}:"); traceback.print_exc()
            if fails >= 5: sys.exit(1)
        finally:
            env.undo()
    if r % 5 == 4: print(f"round {r + 1}/{first + rounds} done, {fails} failures, {time.time() - t0:.0f} s", flush=True)
np.random.default_rng = real_rng
print("soak finished:", rounds, "rounds,", fails, "failures")
