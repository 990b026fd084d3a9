"""libmovba's HOST side under ThreadSanitizer on the CPU: api.cpp (upload with its helper thread and the shared copy stream, the
LM loop polling the device's progress, park / resume for the direct solver, batched runs, the stop flag, the watchdog),
structure.cpp, dense_plan.cpp and pcg_plan.cpp compiled against a stand-in HIP runtime whose streams are threads
(tests/hipstub/: test infrastructure, nothing of it ships), driven from several threads.  The fake device follows the LM
controller's protocol and can park a solve or stop making progress; it computes no bundle adjustment."""
import os
import subprocess

from conftest import ROOT

STUB = os.path.join(ROOT, "tests", "hipstub")


def test_host_state_machine_is_race_free_and_the_watchdog_fires_on_a_stalled_device():
    subprocess.check_call(["make", "-C", STUB, "-s"])
    env = dict(os.environ, TSAN_OPTIONS="halt_on_error=0 exitcode=66")
    p = subprocess.run([os.path.join(STUB, "host_tsan_driver")], env=env, capture_output=True, text=True, timeout=600)
    assert "WARNING: ThreadSanitizer" not in p.stderr, p.stderr[:4000]
    assert p.returncode == 0 and p.stdout.strip().endswith("HOST-TSAN OK"), p.stderr[-2000:]
    # scenario 5 of the driver: a device that stops making progress gives the call back through the watchdog
    assert "device made no progress for 200 ms" in p.stderr


def test_host_side_under_address_and_undefined_behaviour_sanitizers():
    """The same driver (uploads from several threads, batches, park / resume, the stop flag, the watchdog) built with
    -fsanitize=address,undefined: the host's layout carving, staging-buffer packing and schedule building overrun nothing."""
    subprocess.check_call(["make", "-C", STUB, "-s", "host_asan_driver"])
    env = dict(os.environ, ASAN_OPTIONS="detect_leaks=0 abort_on_error=0 exitcode=67", UBSAN_OPTIONS="print_stacktrace=1")
    p = subprocess.run([os.path.join(STUB, "host_asan_driver")], env=env, capture_output=True, text=True, timeout=600)
    assert "ERROR: AddressSanitizer" not in p.stderr and "runtime error:" not in p.stderr, p.stderr[:4000]
    assert p.returncode == 0 and p.stdout.strip().endswith("HOST-TSAN OK"), p.stderr[-2000:]


def test_adapter_under_address_and_undefined_behaviour_sanitizers(tmp_path):
    """mov-slam_amd/host/Optimizer.cc over the mock map classes, with libmovba's host side and the fake device linked in
    (tests/hipstub: adapter_asan), built with -fsanitize=address,undefined: LocalBundleAdjustment on a monocular window, a
    stereo one with a camera per keyframe and one with a bad observer, GlobalBundleAdjustemnt and PoseOptimization.  The fake
    device returns the start poses: what is checked is every pointer the adapter follows and every buffer it sizes."""
    import struct
    import sys
    import numpy as np
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from movba import synth
    import test_gpu_adapter as ga                       # (the window file format of the adapter test binary)
    subprocess.check_call(["make", "-C", STUB, "-s", "adapter_asan"])
    exe = os.path.join(STUB, "adapter_asan")
    env = dict(os.environ, ASAN_OPTIONS="detect_leaks=0 exitcode=67", UBSAN_OPTIONS="print_stacktrace=1")

    def run(mode, fin):
        p = subprocess.run([exe, mode, fin, str(tmp_path / "out.bin")], env=env, capture_output=True, text=True, timeout=300)
        assert "ERROR: AddressSanitizer" not in p.stderr and "runtime error:" not in p.stderr, p.stderr[:4000]
        assert p.returncode == 0, p.stderr[-2000:]

    cases = [("lba", synth.cfg("small"), -1), ("lba", synth.cfg("cfg2"), -1),
             ("lba", synth.mixed_cameras(synth.make_window(6, 2, 150, seed=43, run_lo=2, run_hi=5, stereo_frac=0.7), seed=44), -1),
             ("lba", synth.cfg("small"), 1),
             ("gba", synth.cfg("small"), -1)]
    for k, (mode, w, bad) in enumerate(cases):
        w.poses = ga._f32_pose(w.poses)
        fin = str(tmp_path / f"w{k}.bin")
        ga._write_window(fin, w, bad_kf=bad)
        run(mode, fin)
    f = synth.make_frame(n=300, seed=5)
    fin = str(tmp_path / "pose.bin")
    with open(fin, "wb") as fh:
        fh.write(struct.pack("2i", 300, 0))
        fh.write(np.ascontiguousarray(f["Xw"], np.float64).tobytes())
        fh.write(np.ascontiguousarray(f["obs"], np.float64).tobytes())
        fh.write(np.ascontiguousarray(f["pose0"], np.float64).tobytes())
    run("pose", fin)
