"""View split launched as a chain of HIP-graph SEGMENTS (parallel.SegmentedGraph) == its eager form, bit for bit, on every
rank (graph capture as the default of the sharded modes once a shared-GPU gloo run shows replay ==
eager).  Real `torch.distributed` ranks under torchrun, all on the one GPU of the box, gloo transport (host-staged
exchanges between the segments): tools/view_split_segments_check.py."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.parametrize("world", [3, 4], ids=["3_shards_both_halves", "2_halves_x_2_shards"])
def test_segmented_replay_equals_eager(gpu, world):
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
    env.update({"DD_BENCH_SHARE_GPU": "1", "DD_BENCH_BACKEND": "gloo", "OMP_NUM_THREADS": "4"})
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(world),
           "--master-addr", "127.0.0.1", "--master-port", str(29700 + world),
           os.path.join(ROOT, "tools", "view_split_segments_check.py")]
    r = subprocess.run(cmd, env=env, cwd=ROOT, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-2000:]
    line = [ln for ln in r.stdout.splitlines() if ln.startswith("{")][-1]
    out = json.loads(line)
    print(out)
    assert out["world"] == world and out["segments"] == 17 and out["exchanges"] == 16      # 16 UNet transformer blocks
    assert out["bitwise_equal_all_ranks"] is True
