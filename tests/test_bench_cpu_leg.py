"""bench.py's cpu_baseline leg (the only place of the bench that touches the oracle) on CPU: a small graph dumped by
the oracle, the "device answers" played by the oracle's own answers, so parity must come out clean and the timing
fields must be filled for both arithmetic variants."""
import numpy as np

from conftest import uniform


def test_cpu_baseline_leg_reports_timing_and_parity(oracle, tmp_path):
    import bench
    n, d, k, ef, nq = 3000, 32, 10, 48, 300
    X = uniform(n, d, 3)
    o = oracle.OracleHnsw(16, n, 16, 100, "DistL2")
    o.insert_batch(X)
    o.file_dump(tmp_path, "leg")
    Q = uniform(nq, d, 4)
    ref = o.parallel_search(Q, k, ef)
    st = np.zeros((nq, 8), np.int64)
    st[5, 3] = 3   # one query "answered by the literal heaps", one "flagged"
    st[9, 3] = 2
    cpu, parity = bench.cpu_baseline_leg(str(tmp_path), "leg", "DistL2", Q, k, ef, ref.ids.astype(np.int64), ref.dists.copy(),
                                         st, ref.counts.astype(np.int32), cpu_seconds=0.5)
    assert cpu["kind"] == "port" and cpu["unit"] == "queries/s" and cpu["value"] > 0 and cpu["cores"] >= 1
    assert set(cpu["by_threads_simd_order"]) <= set(cpu["by_threads"]) and cpu["usable_cpus"] >= 1  # (the two best thread counts again in the SIMD order)
    assert parity["queries_checked"] == nq
    assert parity["tie_free_ids_identical"] and parity["tie_free_f32_distance_bits_identical"]
    assert parity["all_ids_identical"] and parity["all_f32_distance_bits_identical"]
    assert parity["queries_whose_answer_depends_on_heap_order"] == 2 and parity["resolved_with_literal_heaps"] == 1
    assert parity["heap_order_queries_ids_identical"] == 2 and parity["heap_order_queries_distance_bits_identical"] == 2
    assert cpu["protocol"].startswith("1 warm-up") and cpu["flat_avx"]["value"] > 0 and cpu["flat_avx"]["ids_agreeing_with_the_port"] > 0.98
    # a corrupted device answer must show up
    bad = ref.ids.astype(np.int64).copy()
    bad[0, 0] ^= 1
    _, parity2 = bench.cpu_baseline_leg(str(tmp_path), "leg", "DistL2", Q, k, ef, bad, ref.dists.copy(), st,
                                        ref.counts.astype(np.int32), cpu_seconds=0.2)
    assert not parity2["tie_free_ids_identical"] and not parity2["all_ids_identical"]
