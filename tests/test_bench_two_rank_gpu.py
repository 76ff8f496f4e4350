"""bench.py's N > 1 path on real hardware, as far as a one-GPU box allows: two ranks launched the way the driver launches them
(torch.distributed.run, 127.0.0.1 rendezvous), both folded onto cuda:0 (parallel.local_device) with gloo carrying the barrier, the
max-over-ranks timing and the end-of-batch all-reduce of the episodic returns (SGW_DIST_BACKEND: the rehearsal switch; two processes
cannot share one device under RCCL).  Checks the contract fields of the ONE JSON line rank 0 prints and that the shards do not overlap:
twice the envs, twice the finished episodes of one shard's rate."""
import json
import os
import socket
import subprocess
import sys

import pytest

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
  with socket.socket() as s:
    s.bind(("127.0.0.1", 0))
    return s.getsockname()[1]


@pytest.mark.gpu
def test_two_rank_bench_line():
  env = dict(os.environ, SGW_DIST_BACKEND="gloo", MASTER_ADDR="127.0.0.1")
  cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
         "--master-port", str(_free_port()), os.path.join(REPO, "bench.py"), "--gpus", "2", "--steps", "20", "--warmup", "5",
         "--no-cpu-baseline", "--no-fused", "--min-seconds", "0.1"]
  p = subprocess.run(cmd, cwd=REPO, env=env, capture_output=True, text=True, timeout=900)
  assert p.returncode == 0, p.stderr[-3000:]
  lines = [l for l in p.stdout.splitlines() if l.startswith("{")]
  assert len(lines) == 1, p.stdout[-2000:]                       # rank 0 alone prints
  d = json.loads(lines[0])
  assert d["n_gpus"] == 2 and d["scaling"] == "weak" and d["higher_is_better"] is True
  assert d["steps"] == 20 and d["warmup"] == 5 and d["data"] == "synthetic" and d["dtype"] == "f64"
  assert d["config"]["envs_per_gpu"] == 65536 and "dp2" in d["config"]["sharding"]
  assert d["value"] > 0 and abs(d["value"] - 2 * 65536 / (d["ms_per_step"] * 1e-3)) < 1e-6 * d["value"]     # whole-job aggregate
  ret = d["returns"]["island_navigation_ex"]
  assert ret["episodes_finished"] > 0 and len(ret["mean_episode_return"]) == 10
  assert d["roofline"]["bound"] == "hbm" and 0 < d["roofline"]["frac"] < 1
