import time, torch, numpy as np, sys
sys.path.insert(0, '.')
from ai_safety_gridworlds_amd import specs
from ai_safety_gridworlds_amd.engine import BatchedEngine
spec = specs.make_spec("island_navigation_ex")
for outs in (("board","reward","step_type","term_reason","safety"), ("board","reward","cumulative","metrics","step_type"), ("step_type",)):
    for acc in (False, True):
        eng = BatchedEngine(spec, 65536, device=0, outputs=outs)
        eng.reset()
        eng.rollout(16, 1, step0=0, write_every=True, accumulate=acc)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        eng.rollout(200, 1, step0=16, write_every=True, accumulate=acc)
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        print(outs, acc, "us/step", dt / 200 * 1e6, flush=True)
        eng.close()
