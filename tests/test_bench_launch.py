"""bench.py's own launcher: ``python bench.py --gpus N`` starts N ranks itself, fails fast when the
node has fewer devices, and reports the world size the communicator saw.  Runs the real bench code path on the CPU
(``--device cpu``): native TcpStore rendezvous, C++ reducer over the host backend, barrier / MAX-reduce timing."""
import json
import os
import subprocess
import sys

from distributed_pytorch_amd import launch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _bench(*argv, env=None, timeout=600):
    e = dict(os.environ)
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        e.pop(k, None)
    e.update(env or {})
    return subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + list(argv), capture_output=True,
                          text=True, timeout=timeout, env=e, cwd=ROOT)


def test_bench_self_launches_two_ranks_on_cpu():
    r = _bench("--gpus", "2", "--device", "cpu", "--arch", "resnet18", "--image-size", "32", "--num-classes", "10",
               "--batch-size", "4", "--steps", "3", "--warmup", "2", env={"DPA_NPROC_PER_NODE": "2"})
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout                      # rank 0 prints the ONE json line
    j = json.loads(lines[0])
    assert j["config"]["ranks"] == 2 and j["config"]["global_batch"] == 8 and j["config"]["parallelism"] == "dp2"
    comm = j["config"]["comm"]
    assert comm["reducer"] == "c++" and comm["ranks"] == 2 and comm["process_group"] == "native"
    assert comm["buckets"] >= 1 and comm["bytes_per_step"] > 4 * 11_000_000 and comm["collectives"] >= 5
    assert j["steps"] == 3 and j["warmup"] == 2 and j["value"] > 0 and j["final_loss"] is not None


def test_bench_under_torchrun_exactly_as_the_driver_launches_it():
    """``python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P bench.py
    --gpus N ...``: ranks come from the environment, MASTER_PORT belongs to torchrun's agent store (ours takes an
    ephemeral port published through it), rank 0 prints the one JSON line."""
    e = dict(os.environ)
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        e.pop(k, None)
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2",
                        "--master-addr", "127.0.0.1", "--master-port", str(launch.free_port()),
                        os.path.join(ROOT, "bench.py"), "--gpus", "2", "--device", "cpu", "--arch", "resnet18",
                        "--image-size", "32", "--num-classes", "10", "--batch-size", "4", "--steps", "3", "--warmup", "2"],
                       capture_output=True, text=True, timeout=600, env=e, cwd=ROOT)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout
    j = json.loads(lines[0])
    assert j["config"]["ranks"] == 2 and j["config"]["parallelism"] == "dp2" and j["steps"] == 3
    assert j["config"]["comm"]["process_group"] == "native" and j["config"]["comm"]["reducer"] == "c++"


def test_bench_fails_fast_when_the_node_has_too_few_devices():
    r = _bench("--gpus", "2", "--device", "cpu", env={"DPA_NPROC_PER_NODE": "1"}, timeout=120)
    assert r.returncode == 2 and "--gpus 2 requested but this node has 1" in r.stderr
    # a launcher that provides fewer ranks than --gpus is refused as well (round 1 silently ran one GPU)
    r = _bench("--gpus", "2", "--device", "cpu", env={"WORLD_SIZE": "1", "RANK": "0"}, timeout=120)
    assert r.returncode != 0 and "--gpus 2 but WORLD_SIZE=1" in r.stderr


def test_affinity_plan_follows_the_gpus_numa_node():
    node_cpus = {0: list(range(0, 64)), 1: list(range(64, 128))}.get
    allowed = list(range(128))
    nodes = [0, 0, 0, 0, 1, 1, 1, 1]                       # 8 GPUs, 4 per socket
    plans = [launch.plan_affinity(r, 8, allowed, nodes, node_cpus) for r in range(8)]
    assert plans[0] == list(range(0, 16)) and plans[3] == list(range(48, 64))
    assert plans[4] == list(range(64, 80)) and plans[7] == list(range(112, 128))
    assert sorted(sum(plans, [])) == allowed              # disjoint, complete
    # a cgroup that only allows part of the node: stay inside it
    part = launch.plan_affinity(5, 8, list(range(64, 96)), nodes, node_cpus)
    assert part == list(range(72, 80))
    # unknown topology: even slices of whatever is allowed
    assert launch.plan_affinity(1, 2, [0, 1, 2, 3], [None, None], node_cpus) == [2, 3]
    # interleaved GPU -> node mapping
    inter = [launch.plan_affinity(r, 4, allowed, [0, 1, 0, 1], node_cpus) for r in range(4)]
    assert inter[0] == list(range(0, 32)) and inter[2] == list(range(32, 64)) and inter[1] == list(range(64, 96))
