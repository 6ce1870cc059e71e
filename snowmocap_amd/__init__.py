"""snowmocap_amd -- MI355X-native multi-view triangulation core behind SnowMocap's
`snowvision.triangulation` / `camera.CameraGroup` API.

Importing the package does not load the HIP library; the first call that needs it does, and
fails loudly if libsnowtri.so was not built or no GPU is visible (there is no CPU fallback).
"""
from .camera import Camera, CameraGroup
from .triangulation import (Skew_Ray_Solver, Human_Triangulation, Human_Triangulation_Condense,
                            Human_Triangulation_Smooth, SecondOrderDynamic, skew_ray_solver_batch, smooth_track)
from .util import Load_Config_Json, Check_If_File_Exist
from .batch import BatchTriangulator

__all__ = ["Camera", "CameraGroup", "Skew_Ray_Solver", "Human_Triangulation",
           "Human_Triangulation_Condense", "Human_Triangulation_Smooth", "SecondOrderDynamic",
           "skew_ray_solver_batch", "smooth_track", "Load_Config_Json", "Check_If_File_Exist", "BatchTriangulator"]
