"""Config / path helpers the hot-path callers need (mirror of snowvision/util.py:7-24).

Only the two plumbing helpers main.py calls around the triangulation path live here; video I/O,
brightness and the matplotlib drawing helpers are outside the path.
"""
from __future__ import annotations

import json
import os


def Check_If_File_Exist(file_path):
    """Non-clobbering output name (util.py:7-19): returns (False, path) when `file_path` is free,
    else (False, "<stem>_<n>.<ext>") for the first free n = 0, 1, ...  Like the reference, the
    name must contain exactly one '.' when an alternative has to be derived."""
    if not os.path.isfile(file_path):
        return False, file_path
    stem, ext = file_path.split(".")
    n = 0
    while True:
        candidate = f"{stem}_{n}.{ext}"
        if not os.path.isfile(candidate):
            return False, candidate
        n += 1


def Load_Config_Json(config_path):
    """util.py:21-24: flat JSON dict, no validation, no defaults."""
    with open(config_path, "r") as fh:
        return json.load(fh)
