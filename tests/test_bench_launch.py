"""bench.py's own launcher, without a GPU: `python bench.py --gpus N` with no launcher around it starts
`python -m torch.distributed.run --standalone --local-addr 127.0.0.1 --nnodes=1 --nproc-per-node N bench.py <same arguments>` as a CHILD process on the loopback
interface, hands back the child's exit code, and does so before anything of torch or libhns is touched."""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_self_launch_command_and_exit_code(monkeypatch):
    sys.path.insert(0, ROOT)
    import bench

    seen = {}

    class Done:
        returncode = 7

    def fake_run(cmd, env=None, **kw):
        seen["cmd"], seen["env"] = cmd, env
        return Done()

    monkeypatch.setattr(subprocess, "run", fake_run)
    monkeypatch.setattr(sys, "argv", ["bench.py", "--gpus", "4", "--steps", "5", "--warmup", "2", "--config", "plume1024", "--partition"])
    monkeypatch.delenv("WORLD_SIZE", raising=False)
    args = bench.parse()
    assert bench.self_launch(args) == 7
    cmd = seen["cmd"]
    assert cmd[:3] == [sys.executable, "-m", "torch.distributed.run"]
    assert "--nnodes=1" in cmd and "--nproc-per-node=4" in cmd
    assert "--standalone" in cmd and cmd[cmd.index("--local-addr") + 1] == "127.0.0.1" and "--master-port" not in cmd  # (torchrun binds its own port)
    tail = cmd[cmd.index(os.path.join(ROOT, "bench.py")) + 1:]
    assert tail == ["--gpus", "4", "--steps", "5", "--warmup", "2", "--config", "plume1024", "--partition"]
    assert seen["env"]["HSA_ENABLE_IPC_MODE_LEGACY"] == os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0")


def test_parent_launches_before_it_imports_torch():
    """The parent of `bench.py --gpus 2` must not initialise the GPU: it never gets as far as importing torch or loading libhns."""
    code = (
        "import sys, subprocess, runpy\n"
        "subprocess.run = lambda cmd, env=None, **kw: type('R', (), {'returncode': 0})()\n"
        f"sys.argv = [{os.path.join(ROOT, 'bench.py')!r}, '--gpus', '2']\n"
        "try:\n"
        f"    runpy.run_path({os.path.join(ROOT, 'bench.py')!r}, run_name='__main__')\n"
        "except SystemExit as e:\n"
        "    assert e.code == 0, e.code\n"
        "assert 'torch' not in sys.modules and 'hnanosolver_amd' not in sys.modules, sorted(m for m in sys.modules if m.startswith(('torch', 'hnano')))\n"
        "print('ok')\n"
    )
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    p = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, env=env, cwd=ROOT)
    assert p.returncode == 0 and p.stdout.strip() == "ok", p.stdout + p.stderr


def test_one_sided_sweeps_per_exchange_by_rank_size():
    from hnanosolver_amd.dist import SlabBench

    assert SlabBench._one_sided_k(65944, 8) == 2  # BASELINE config 5 in 8 ranges: the temporally blocked chained sweep
    assert SlabBench._one_sided_k(2 * 32768, 2) == 2
    assert SlabBench._one_sided_k(512, 2) == 1 and SlabBench._one_sided_k(4800, 8) == 1  # 600 leaves per rank and fewer: one-leaf blocks, one iteration per chained launch


def test_watchdog_cuts_a_call_that_blocks_in_c():
    """the strong-scaling records of `bench.py --gpus N` are guarded by a watchdog THREAD, because a hung collective blocks the main thread inside C where no Python
    signal handler runs. Emulated without a GPU: libc's sleep() through ctypes (GIL released, never returns to the bytecode loop in time). The watchdog's last words come
    out, the process exits 0 at once, and what follows the blocked call never runs."""
    import time

    code = (
        "import sys, ctypes\n"
        f"sys.path.insert(0, {ROOT!r})\n"
        "import bench\n"
        "def last_words():\n"
        "    sys.stdout.write('{\"abandoned\": true}\\n'); sys.stdout.flush()\n"
        "dog = bench.Watchdog(1.0, last_words)\n"
        "ctypes.CDLL(None).sleep(60)\n"
        "dog.cancel()\n"
        "print('not reached')\n"
    )
    t0 = time.time()
    p = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, cwd=ROOT, timeout=50)
    assert p.returncode == 0 and p.stdout.strip() == '{"abandoned": true}', p.stdout + p.stderr
    assert time.time() - t0 < 30
    # and a record that completes in time is not disturbed
    code2 = f"import sys\nsys.path.insert(0, {ROOT!r})\nimport bench, time\ndog = bench.Watchdog(5.0, lambda: print('fired'))\ntime.sleep(0.1)\ndog.cancel()\nprint('done')\n"
    p = subprocess.run([sys.executable, "-c", code2], capture_output=True, text=True, cwd=ROOT, timeout=50)
    assert p.returncode == 0 and p.stdout.strip() == "done", p.stdout + p.stderr
