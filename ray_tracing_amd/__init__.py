"""MI355X-native path-tracing backend for the `ray_tracer` crate's render loop.

`ray_tracing_amd.api` binds the C ABI of include/rt_amd.h; `ray_tracing_amd.scenes` holds the
benchmark scene descriptions.  Importing fails loudly if the HIP library has not been built.
"""
from . import api, scenes  # noqa: F401
from .api import Comm, Context, FramePipeline, Mesh, RtError, Scene, SceneDescription, comm_unique_id  # noqa: F401
