"""k_gemm_dma's two staging forms -- buffer descriptors (round 6, shipped) and 64-bit addresses (rounds 1-5) -- give bit-identical
outputs: a fixed-seed run of tools/fuzz_gemm_stage.py (random dense / column-window / 3x3-convolution / GEGLU launches, ragged row
counts, every epilogue, fp16 and bf16, with and without K split), and the folded-LayerNorm instantiations, which the sweep's hook does
not reach, by running the engine's LayerNorm-fold test under both forms."""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_buffer_descriptor_staging_is_bit_identical_to_the_address_form():
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "fuzz_gemm_stage.py"), "120", "6"], capture_output=True, text=True,
                       timeout=1200)
    assert r.returncode == 0, (r.stdout[-2000:], r.stderr[-2000:])
    assert "120 cases, buffer-descriptor staging bit-identical" in r.stdout, r.stdout[-500:]


@pytest.mark.parametrize("stage", ["0", "1"])
def test_layernorm_fold_under_both_staging_forms(stage):
    env = dict(os.environ, DH_GEMM_STAGE=stage)
    r = subprocess.run([sys.executable, "-m", "pytest", os.path.join(ROOT, "tests", "test_unet_kernels_gpu.py"), "-q", "-m", "gpu", "-k",
                        "layernorm_fold or geglu_epilogues"], env=env, capture_output=True, text=True, timeout=1200, cwd=ROOT)
    assert r.returncode == 0, r.stdout[-3000:]
