"""``d_infer.py`` of the reference is ``infer.py`` plus a root depth fed to the camera maths as
``depth_refine`` (d_infer.py:355,:438-440,:1275-1277 -> renderer.py:47-52).  ``estimate_from_rgb``
here already takes ``depth_refine``; the RootNet regressor that produces the depth is listed as
"next" in SURVEY.md 8(f) and is not part of this build yet, so the caller supplies the depth."""
from .infer import *  # noqa: F401,F403
from .infer import hamer_inference  # noqa: F401
