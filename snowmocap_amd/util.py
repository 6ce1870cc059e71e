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


def Load_Video(video_name):
    """util.py:34-41 (video I/O, outside the triangulation path): OpenCV capture + length + image size.
    OpenCV is imported lazily: the triangulation core itself never needs it."""
    import cv2
    cap = cv2.VideoCapture(video_name)
    return {"cap": cap,
            "length": int(cap.get(cv2.CAP_PROP_FRAME_COUNT)),
            "image_size": (int(cap.get(cv2.CAP_PROP_FRAME_WIDTH)), int(cap.get(cv2.CAP_PROP_FRAME_HEIGHT)))}


def _display_only(name):
    def stub(*_a, **_k):
        raise NotImplementedError(f"{name} is matplotlib display code outside the triangulation path; "
                                  "import it from the original snowvision.util")
    stub.__name__ = name
    return stub


Draw_Camera_Group = _display_only("Draw_Camera_Group")
Draw_Skeleton = _display_only("Draw_Skeleton")
