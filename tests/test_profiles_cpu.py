"""CPU: the committed evidence must describe the committed kernels.  bench.py reads `roofline.traffic` from the newest
profiles/*_traffic.json (PMC passes collected by tools/collect_profiles.sh); if a kernel source changed after that file was
committed, the number in the bench line is stale and nothing else would notice."""
import glob
import os
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _git(*args):
    return subprocess.run(["git", "-C", ROOT] + list(args), capture_output=True, text=True, timeout=60)


def test_traffic_profile_is_not_older_than_the_kernels_it_describes():
    if not os.path.isdir(os.path.join(ROOT, ".git")) or _git("rev-parse", "HEAD").returncode != 0:
        pytest.skip("not a git checkout (GPU box snapshot)")
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", "*_traffic.json")))
    assert files, "no profiles/*_traffic.json"
    newest = os.path.relpath(files[-1], ROOT)
    t_prof = _git("log", "-1", "--format=%ct", "--", newest).stdout.strip()
    assert t_prof, "%s is not committed" % newest
    # the kernels of the default bench path whose HBM traffic the file reports (conv class + warp_costvol)
    kernels = ["back2future_amd/csrc/" + f for f in ("b2f_wino6.hip", "b2f_wino4.hip", "b2f_wino.hip", "b2f_conv.hip", "b2f_conv16.hip", "b2f_head.hip", "b2f_convb.hip", "b2f_s2b.hip", "b2f_w1b.hip", "b2f_corr.hip",
                                                     "b2f_corr5.hip", "b2f_corr5_loop.inc", "b2f_glue.hip")]
    t_k = _git("log", "-1", "--format=%ct", "--", *kernels).stdout.strip()
    assert int(t_prof) >= int(t_k), ("%s was committed before the last change to the kernels it describes: re-run "
                                     "tools/collect_profiles.sh on the GPU box and commit the new profiles" % newest)
