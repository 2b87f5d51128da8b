"""The launcher's host-side decisions (eigen_amd/csrc/gemm_launch.hpp), queried through eigamd_gemm_plan: no GPU is
needed and nothing is launched.  Pinned here are the properties the design relies on, not the model's constants:
products of many tiles run whole with the large tile; products of few tiles are cut up one way or another; a product
that overwrites one of its operands keeps the large tile (the triangular solves rely on one tile covering the block);
the pieces of a reduction stay within the limits the partial-product buffers are sized for; the environment overrides."""
import ctypes
import os

import pytest

import eigen_amd
from eigen_amd import _lib

KINDS = [eigen_amd.F32, eigen_amd.F64, eigen_amd.C32, eigen_amd.C64]
VARIABLE = {eigen_amd.F64: "EIGAMD_F64_TILE", eigen_amd.F32: "EIGAMD_F32_TILE", eigen_amd.C64: "EIGAMD_C64_TILE",
            eigen_amd.C32: "EIGAMD_C32_TILE"}


def plan(kind, m, n, k, in_place=0):
    out = (ctypes.c_int * 5)()
    rc = _lib.lib.eigamd_gemm_plan(kind, m, n, k, in_place, out)
    assert rc == 0, _lib.lib.eigamd_last_error()
    return dict(small=out[0], pieces=out[1], tail_pieces=out[2], bm=out[3], bn=out[4])


def large_tile(kind):
    bm, bn, bk = ctypes.c_int(), ctypes.c_int(), ctypes.c_int()
    _lib.lib.eigamd_tile_shape(kind, ctypes.byref(bm), ctypes.byref(bn), ctypes.byref(bk))
    return bm.value, bn.value, bk.value


@pytest.mark.parametrize("kind", KINDS)
def test_products_of_many_tiles_run_whole_with_the_large_tile(kind):
    bm, bn, _ = large_tile(kind)
    for shape in [(16384, 16384, 16384), (8192, 8192, 8192), (31744, 31744, 1024), (31744, 31744, 256)]:
        p = plan(kind, *shape)
        assert p == dict(small=0, pieces=1, tail_pieces=0, bm=bm, bn=bn), (shape, p)


@pytest.mark.parametrize("kind", KINDS)
def test_products_of_few_tiles_are_cut_up(kind):
    bm, bn, bk = large_tile(kind)
    for shape in [(512, 512, 512), (256, 256, 4096), (1024, 1024, 1024), (512, 512, 8192)]:
        p = plan(kind, *shape)
        assert p["small"] == 1 or p["pieces"] > 1, (shape, p)
        if p["small"]:
            assert p["bm"] * p["bn"] * 4 == bm * bn      # a quarter of the large tile
        assert 1 <= p["pieces"] <= 16
        if p["pieces"] > 1:
            assert shape[2] / p["pieces"] >= 4 * bk - bk   # at least four k-tiles per piece (up to rounding)
            assert shape[0] * shape[1] * p["pieces"] * {0: 4, 1: 8, 2: 8, 3: 16}[kind] <= 256 << 20


@pytest.mark.parametrize("kind", KINDS)
def test_in_place_products_keep_the_large_tile(kind):
    bm, bn, _ = large_tile(kind)
    for shape in [(bm, 64, bm), (bm, 1024, bm), (bm, 8192, bm), (128, 4096, 128), (256, 256, 256)]:
        p = plan(kind, *shape, in_place=1)
        assert p["small"] == 0 and (p["bm"], p["bn"]) == (bm, bn), (shape, p)


@pytest.mark.parametrize("kind", KINDS)
def test_environment_fixes_the_tile(kind):
    var = VARIABLE[kind]
    old = os.environ.get(var)
    try:
        for shape in [(512, 512, 512), (4096, 4096, 4096), (2560, 2560, 2560)]:
            os.environ[var] = "1"
            assert plan(kind, *shape)["small"] == 0
            os.environ[var] = "2"
            assert plan(kind, *shape)["small"] == 1
    finally:
        if old is None:
            os.environ.pop(var, None)
        else:
            os.environ[var] = old


def test_a_mostly_empty_last_round_gets_a_tail():
    # dgemm 5120^3 with the large tile: 800 tiles, 3.1 rounds of 256 -- whole rounds plus a tail whose reduction is cut
    old = os.environ.get("EIGAMD_F64_TILE")
    os.environ["EIGAMD_F64_TILE"] = "1"
    try:
        p = plan(eigen_amd.F64, 5120, 5120, 5120)
        assert p["pieces"] == 1 and p["tail_pieces"] > 1, p
        # exactly two rounds: nothing to cut
        p = plan(eigen_amd.F64, 4096, 4096, 4096)
        assert p["pieces"] == 1 and p["tail_pieces"] == 0, p
    finally:
        if old is None:
            os.environ.pop("EIGAMD_F64_TILE", None)
        else:
            os.environ["EIGAMD_F64_TILE"] = old


def test_invalid_queries():
    out = (ctypes.c_int * 5)()
    assert _lib.lib.eigamd_gemm_plan(7, 10, 10, 10, 0, out) != 0
    assert _lib.lib.eigamd_gemm_plan(1, 0, 10, 10, 0, out) != 0
    assert _lib.lib.eigamd_gemm_plan(1, 10, 10, 10, 0, None) != 0
