"""Frame time and tail-kernel share for a few switch-over thresholds (RT_TAIL_PATHS is read once per process,
so this script is run once per value)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import ray_tracing_amd as rt
from ray_tracing_amd import scenes
desc = scenes.path_scene(levels=9)
scene = rt.Scene(desc); ctx = rt.Context(0); ctx.upload(scene)
acc = torch.zeros(1080, 1920, 3, device="cuda")
for _ in range(2):
    _, st = ctx.render(1920, 1080, 64, 16, 0, 0, flags=rt.api.RENDER_TIME_STAGES, out=acc.data_ptr())
print(f"RT_TAIL_PATHS={os.environ.get('RT_TAIL_PATHS','default')}: total {st.ms_total:.1f} ms ext {st.ms_extend:.1f} shadow {st.ms_shadow:.1f} shade {st.ms_shade:.1f} "
      f"compact {st.ms_compact:.1f} gen {st.ms_generate:.1f} tail {st.ms_tail:.2f} (paths {st.paths_tail}, rays {st.rays_closest_tail}+{st.rays_shadow_tail}) "
      f"iterations {st.iterations} launches_extend {st.launches_extend}")
