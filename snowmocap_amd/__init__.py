"""snowmocap_amd -- MI355X-native multi-view triangulation core behind SnowMocap's
`snowvision.triangulation` / `camera.CameraGroup` API.

Importing the package does not load the HIP library; the first call that needs it does, and
fails loudly if libsnowtri.so was not built or no GPU is visible (there is no CPU fallback).
"""
from .camera import Camera, CameraGroup
from .triangulation import (Skew_Ray_Solver, Human_Triangulation, Human_Triangulation_Condense,
                            Human_Triangulation_Smooth, SecondOrderDynamic, skew_ray_solver_batch, smooth_track)
from .util import Load_Config_Json, Check_If_File_Exist, Load_Video, Draw_Camera_Group, Draw_Skeleton
from .blender import (Human_Triangulation_Blender, Human_Triangulation_Blender_Smooth,
                      Human_Triangulation_To_Blender_Result, save_blender_result)
from .batch import BatchTriangulator
from .pipeline import ShardedTrackPipeline, TrackPipeline

__all__ = ["Camera", "CameraGroup", "Skew_Ray_Solver", "Human_Triangulation",
           "Human_Triangulation_Condense", "Human_Triangulation_Smooth", "SecondOrderDynamic",
           "skew_ray_solver_batch", "smooth_track", "Load_Config_Json", "Check_If_File_Exist", "Load_Video", "Draw_Camera_Group", "Draw_Skeleton", "BatchTriangulator", "TrackPipeline", "ShardedTrackPipeline", "Human_Triangulation_Blender",
           "Human_Triangulation_Blender_Smooth", "Human_Triangulation_To_Blender_Result", "save_blender_result"]
