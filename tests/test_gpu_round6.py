"""Round-6 GPU tests: BASELINE config 5's shape beyond the Infinity Cache at full size against the oracle."""
import pytest

from test_gpu_round3 import _full_size

pytestmark = pytest.mark.gpu


def test_full_size_parity_mnist784_hbm_240k_x_784(native, oracle, tmp_path, knob):
    """bench.py --config mnist784_hbm: 240 000 x 784 (753 MB of vectors: four times BASELINE config 5, out of the 256 MiB Infinity
    Cache), M = 32 (64 ids per row), ef = 200 (4 result slots per lane): GPU-assisted build -> dump -> product and oracle reload the
    same files -> ids, f32 distance bits, p_ids and counts identical, also with every pop from the literal heap and with a visited
    table far too small (src/hnsw.rs:922-1064, :1487-1580)."""
    _full_size(native, oracle, tmp_path, knob, 240_000, 784, 32, 400, "DistL2", 10, 200, 1500, 600, 300)
