"""Mechanical guard (VERDICT r4, next #1b): every kernel of this library that a committed profile names -- the kernels the
bench numbers and roofline fractions are quoted on -- must have been LAUNCHED by a test that compares its results with the
oracle or a reference-generated fixture.

The diagnostic library (tests/_diag/libdclnet_hip_diag.so, same sources as the product, -DDCL_DIAG) counts every launch by
kernel, template arguments included (csrc/common.h: launch census).  This test switches the package to that library, runs
the oracle-comparing tests that mirror the profiled workloads -- the same shapes, so the same template instances and the
same size-dependent kernel choices -- and then reads profiles/<round>_*_kernel_stats.csv: a library kernel named there that
the census has not seen fails the test by name."""
import csv
import ctypes
import glob
import os
import re

import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def kernel_key(name):
    """rocprofv3's / the demangler's kernel name -> 'k_sparse_conv_dma<128, 4, 2, 2, true>' (no return type, namespace or
    parameter list)"""
    n = name.replace("(anonymous namespace)::", "")
    n = re.sub(r"^void ", "", n)
    return n.split("(")[0].strip()


def ours(key):
    return key.startswith("k_") or key == "voxelize_fp_kernel"


def latest_round():
    rounds = sorted({int(m.group(1)) for f in glob.glob(os.path.join(ROOT, "profiles", "r*_kernel_stats.csv"))
                     for m in [re.match(r"r(\d+)_", os.path.basename(f))] if m})
    return "r%d" % rounds[-1]


def profiled_kernels(rnd):
    """{kernel key: [profile tags]} over the round's kernel-stats summaries"""
    out = {}
    for f in sorted(glob.glob(os.path.join(ROOT, "profiles", "%s_*_kernel_stats.csv" % rnd))):
        tag = os.path.basename(f)[len(rnd) + 1:-len("_kernel_stats.csv")]
        for row in csv.DictReader(open(f)):
            k = kernel_key(row["Name"])
            if ours(k):
                out.setdefault(k, []).append(tag)
    return out


def census(lib):
    lib.dcl_debug_launch_census.restype = ctypes.c_longlong
    lib.dcl_debug_launch_census.argtypes = [ctypes.c_char_p, ctypes.c_longlong]
    need = lib.dcl_debug_launch_census(None, 0)
    buf = ctypes.create_string_buffer(int(need) + 16)
    lib.dcl_debug_launch_census(buf, len(buf))
    seen = {}
    for line in buf.value.decode().splitlines():
        name, cnt = line.rsplit("\t", 1)
        seen[kernel_key(name)] = seen.get(kernel_key(name), 0) + int(cnt)
    return seen


def test_every_profiled_kernel_is_launched_by_an_oracle_comparing_test(request, dcl, oracle, golden_dir):
    import test_crops as TC
    import test_gpu_network as TN
    import test_gpu_ops as TO
    lib = TO.enter_diag(dcl, request)
    lib.dcl_debug_launch_census_reset()
    # -- <round>_stress: BASELINE configs[1], 32 crops of N = 12288 / M = 2048, four of them against the oracle graph; the
    #    ADD-S kernel of the bench's metric leg against the reference's expression
    TN.test_stress_shape_full_batch_properties(dcl, oracle)
    TO.test_add_s_matches_reference_expression(dcl)
    #    ... and every tile shape / K-split of the GEMM core (the cost model picks among them by row count) against float64
    TO.test_own_gemm_core_matches_float64_on_every_tile_shape(request, dcl)
    TO.test_own_gemm_core_at_the_benchmarked_row_counts(dcl)
    # -- <round>_ref: 32 crops of N = M = 1024 the way a default Network runs them, four against the oracle graph
    TN.batch_of_reference_shape_crops_vs_oracle(dcl, oracle, 32, "default")
    TN.batch_of_reference_shape_crops_vs_oracle(dcl, oracle, 32, "launch by launch")
    # -- <round>_stream_b1_graph: one LineMOD-shaped crop (5 mm voxels) as a whole-forward graph replay, and the stress crop
    TN.test_forward_matches_oracle_graph(dcl, oracle, 1, 1024, 1024, 0.005)
    TN.test_forward_matches_oracle_graph(dcl, oracle, 1, 12288, 2048, 0.006)
    # -- <round>_primitives: bench.py's primitive shapes
    TO.test_ball_query_and_group_points_at_the_benchmarked_shape(dcl, oracle)
    TO.test_batched_three_nn_and_knn1_bucketed_search_is_exact(request, dcl, oracle, "big")
    TO.test_batched_nn_search_shapes(dcl, oracle, 32, 12288, 2048)
    TO.test_fps_bit_exact_with_ties(dcl, oracle, 12288, 64)
    TO.test_group_and_gather_bit_exact(dcl, oracle)
    TO.test_knn_three_nn_three_interpolate_batched(dcl, oracle)
    # -- <round>_refiner_loop, <round>_crop_builder: reference-generated fixtures
    TN.test_refiner_matches_reference_golden(dcl, golden_dir)
    for graphed in (False, True):
        TN.test_reference_shape_and_stage2_chain_match_reference_golden(dcl, golden_dir, graphed)
    TC.test_device_builders_equal_the_reference_loaders(dcl)
    TC.test_device_crop_builder_bit_exact(dcl, 14, dict(n_obj=6, tiny=5))
    TC.test_one_launch_crop_voxelisation_equals_the_general_device_op(dcl, 6, 1024, 64)
    # -- <round>_conv_layers: the conv / pool ops called layer by layer on real active sets (32 crops)
    TN.test_backbone_levels_and_indices_bit_exact(dcl, oracle)
    TO.test_conv_layers_of_a_32_crop_batch_op_by_op_match_the_oracle(dcl, oracle)
    seen = census(lib)
    rnd = latest_round()
    want = profiled_kernels(rnd)
    assert len(want) >= 30, "profiles/%s_*_kernel_stats.csv name suspiciously few kernels of this library" % rnd
    missing = {k: tags for k, tags in want.items() if seen.get(k, 0) == 0}
    assert not missing, "kernels in profiles/%s_* that no oracle-comparing test above launched: %s" % (rnd, missing)
