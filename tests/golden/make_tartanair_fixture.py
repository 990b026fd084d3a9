#!/usr/bin/env python3
"""Generates tests/golden/tartanair_sample_trajectories.npz in the BUILD container (needs /root/reference).

Inputs: the trajectory pair the reference commits, evaluation/tartanair_eval/evaluation/{pose_est,pose_gt}.txt.
Expected values: produced by the reference's OWN evaluator (tartanair_evaluator.py:19-71, imported from where it lies,
run in a scratch directory because it writes results.png) on that pair, scale=True as its __main__ does for the
monocular track.  Only data travels: the two arrays and the numbers the evaluator returned.

    python tests/golden/make_tartanair_fixture.py
"""
import contextlib
import io
import os
import sys
import tempfile

import numpy as np

sys.dont_write_bytecode = True      # nothing is written under the reference tree

REF = os.environ.get("MOVBA_REFERENCE", "/root/reference")
EVAL = os.path.join(REF, "evaluation", "tartanair_eval", "evaluation")
OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "tartanair_sample_trajectories.npz")


def main():
    gt_path, est_path = os.path.join(EVAL, "pose_gt.txt"), os.path.join(EVAL, "pose_est.txt")
    os.environ.setdefault("MPLBACKEND", "Agg")
    sys.path.insert(0, EVAL)
    import tartanair_evaluator                                   # the reference's evaluator, unmodified
    cwd = os.getcwd()
    with tempfile.TemporaryDirectory() as tmp:
        os.chdir(tmp)
        try:
            with contextlib.redirect_stdout(io.StringIO()):     # it prints whole trajectories
                res = tartanair_evaluator.TartanAirEvaluator().evaluate_one_trajectory(gt_path, est_path, scale=True)
                # the scale the ATE alignment found (evaluator_base.py ATEEvaluator.evaluate -> align): same call, kept
                from evaluator_base import ATEEvaluator
                from trajectory_transform import kitti2tartan
                gt, est = np.loadtxt(gt_path), np.loadtxt(est_path)
                sel_gt = np.array([gt[int(r[0])] for r in est[1:]])
                sel_est = kitti2tartan(np.array([r[1:] for r in est[1:]]))
                import evaluate_ate_scale
                _, _, _, s = evaluate_ate_scale.align(np.matrix(sel_gt[:, :3].T), np.matrix(sel_est[:, :3].T), True)
        finally:
            os.chdir(cwd)
    np.savez_compressed(OUT, pose_est=np.loadtxt(est_path), pose_gt=np.loadtxt(gt_path),
                        expected_ate=float(res["ate_score"]), expected_scale=float(s),
                        expected_n=len(np.loadtxt(est_path)) - 1,
                        note="inputs: files committed by the reference under evaluation/tartanair_eval/evaluation; expected_*: "
                             "returned by the reference's tartanair_evaluator.py (scale=True) run by make_tartanair_fixture.py")
    print("ate_score", res["ate_score"], "scale", s, "->", OUT)


if __name__ == "__main__":
    main()
