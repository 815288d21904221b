"""MI355X-native voxel space-carving back-projection engine.

Drop-in for ONE path of romi/plant-3d-vision: ``plant3dvision.cl.Backprojection`` and the
``Voxels`` task logic that drives it.  Import as ``plant3dvision_amd`` (the directory name
``plant-3d-vision_amd`` is not a Python identifier; ``plant3dvision_amd/`` at the repo root
aliases it).

    from plant3dvision_amd.cl import Backprojection       # plant3dvision/cl.py:47
    from plant3dvision_amd.tasks.cl import voxels_run     # plant3dvision/tasks/cl.py:99
"""
__version__ = "0.1.0"
