"""MI355X-native drop-in for the reference's ``features`` package (DSP feature front-end).

Put ``dsp-speech-recognition_amd/`` on ``sys.path`` ahead of the reference checkout and
``from features import mfcc, delta, to_frames`` / ``from features.endpoint import
basic_endpoint_detection`` (model.py:2,13) resolve here.  Same function names, arguments, defaults
and return types as features/{sigproc,base,endpoint,preprocess}.py; the arithmetic runs in
hand-written HIP kernels (gfx950) behind a ctypes C ABI (include/dsp_frontend.h).

Unlike the reference's ``features/__init__.py`` this import has no side effects (no ./log/
directory, no matplotlib / sklearn import); of the pitch module only the score path is here.
There is no CPU fallback: without the built library or without a GPU every compute call raises.
"""
from .base import *  # noqa: F401,F403
from .sigproc import *  # noqa: F401,F403
from .endpoint import *  # noqa: F401,F403
from .preprocess import *  # noqa: F401,F403
from .pitch import (center_clip, max_pitch, pitch_detect_frame_sr, pitch_detect_sr,  # noqa: F401
                    robust_max_pitch, smooth, window)
from . import base, sigproc, endpoint, preprocess, pitch, batch, pipeline  # noqa: F401
from .batch import FeaturePlan, EndpointPlan  # noqa: F401
from .pipeline import VadMfccPipeline  # noqa: F401
