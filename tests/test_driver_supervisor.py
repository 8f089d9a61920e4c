"""The C driver's --ranks supervisor (nerf_amd/csrc/train_main.c) where no GPU is needed: the process
that forks the ranks never touches the GPU, and a rank that fails must never leave the others blocked
(the reference is single-process, train.c:12-268, and has no counterpart).  On a machine without
enough GPUs every rank fails at its device check, which is exactly the event to survive."""
import glob
import os
import subprocess
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
TRAIN = os.path.join(ROOT, "nerf_amd", "train")
TOO_MANY = "64"  # more ranks than any box has GPUs


def _scene(tmp_path):
    from test_gpu_driver import write_scene
    write_scene(str(tmp_path))
    return ["--data", str(tmp_path), "--images", "4", "--rays", "8", "--samples", "16", "--batches", "2", "--seed", "3",
            "--render-every", "0"]


def _id_files():
    return set(glob.glob("/tmp/nerf_dp_*"))


def test_failing_ranks_end_the_run_with_an_error_and_leave_nothing_behind(tmp_path):
    before = _id_files()
    t0 = time.time()
    p = subprocess.run([TRAIN] + _scene(tmp_path) + ["--ranks", TOO_MANY], cwd=str(tmp_path), capture_output=True,
                       text=True, timeout=60)
    assert p.returncode != 0
    assert "GPUs" in p.stderr and "ending the other ranks" in p.stderr
    assert time.time() - t0 < 30
    assert _id_files() == before, "the communicator-id file of the run must be removed"
    assert not os.path.exists(tmp_path / "x_nerf_model1.bin")


def test_a_blocked_rank_is_ended_when_another_rank_fails(tmp_path):
    """rank 1 blocks before it touches the GPU (the driver's test hook stands in for a rank stuck in
    communicator set-up or a collective); rank 0 fails; the supervisor must end rank 1 and exit
    non-zero instead of waiting for it"""
    before = _id_files()
    env = dict(os.environ, NERF_TRAIN_TEST_HANG_RANK="1")
    t0 = time.time()
    p = subprocess.run([TRAIN] + _scene(tmp_path) + ["--ranks", TOO_MANY], cwd=str(tmp_path), capture_output=True,
                       text=True, timeout=60, env=env)
    assert p.returncode != 0 and "ending the other ranks" in p.stderr
    assert time.time() - t0 < 30
    assert _id_files() == before


def test_stopping_the_supervisor_ends_every_rank(tmp_path):
    """all ranks blocked (hook: rank -1 = every rank): SIGTERM to the supervisor must end them all"""
    env = dict(os.environ, NERF_TRAIN_TEST_HANG_RANK="-1")
    before = _id_files()
    p = subprocess.Popen([TRAIN] + _scene(tmp_path) + ["--ranks", "3"], cwd=str(tmp_path), stdout=subprocess.PIPE,
                         stderr=subprocess.PIPE, text=True, env=env)
    deadline = time.time() + 20
    kids = []
    while time.time() < deadline and len(kids) < 3:
        time.sleep(0.1)
        try:
            kids = [int(x) for x in open(f"/proc/{p.pid}/task/{p.pid}/children").read().split()]
        except OSError:
            kids = []
    assert len(kids) == 3, kids
    p.terminate()
    out, err = p.communicate(timeout=30)
    assert p.returncode != 0 and "stop requested" in err
    for k in kids:
        assert not os.path.exists(f"/proc/{k}"), "a rank outlived the supervisor"
    assert _id_files() == before


def test_ranks_and_virtual_ranks_are_options_of_the_device_side_step(tmp_path):
    for extra in (["--ranks", "2", "--path", "classic"], ["--virtual-ranks", "2", "--path", "classic"]):
        p = subprocess.run([TRAIN] + _scene(tmp_path) + extra, cwd=str(tmp_path), capture_output=True, text=True, timeout=60)
        assert p.returncode != 0 and "--path fused" in p.stderr
    p = subprocess.run([TRAIN] + _scene(tmp_path) + ["--ranks", "2", "--virtual-ranks", "2"], cwd=str(tmp_path),
                       capture_output=True, text=True, timeout=60)
    assert p.returncode != 0 and "Usage" in p.stderr
