"""The LDS budget DESIGN.md section 5 states for the 4-wavefront kernel, from the layout constants themselves (a host program
compiled with hipcc: constexpr only, no GPU needed): the tiles of TWO instances of the reference's problem fit the 160 KB of a CU."""
import os
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")


@pytest.mark.skipif(not (os.path.exists(HIPCC) or shutil.which("hipcc")), reason="hipcc not installed")
def test_two_instances_of_the_reference_problem_fit_one_cu(tmp_path):
    exe = tmp_path / "lds_budget"
    subprocess.run([HIPCC, "-O0", "--offload-arch=gfx950", "-std=c++17", "-I" + os.path.join(ROOT, "include"),
                    "-I" + os.path.join(ROOT, "srbd_horizon_amd", "csrc"), os.path.join(ROOT, "tools", "lds_budget.hip"), "-o", str(exe)],
                   check=True, capture_output=True)
    rows = {l.split()[0]: [int(v) for v in l.split()[1:]] for l in subprocess.run([str(exe)], check=True, capture_output=True, text=True).stdout.splitlines()}
    cu = 160 * 1024
    assert rows["srbd37"][0] <= 51 * 1024 and 3 * rows["srbd37"][0] <= cu          # 52.0 KB = 50.8 KiB (W-free layout; 63.8 KB with the W tile)
    assert 2 * rows["srbd37S"][0] <= cu and 2 * rows["srbd37B"][0] <= cu            # second-order and barrier builds too
    assert 2 * rows["lip30"][0] <= cu
    assert rows["srbd61"][0] <= 147 * 1024                                          # contact_model = 4 (W-free layout, 149.9 KB): one workgroup per CU
    assert rows["srbd61"][0] < rows["srbd61X"][0] <= cu                             # ... and with the 8 user rows (157.2 KB) it still fits
    assert 8 * rows["srbd13"][0] <= cu                                              # one-wave kernel: eight wavefronts per CU
    for name, (nbytes, work, two) in rows.items():
        assert nbytes <= cu, name
