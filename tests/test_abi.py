"""The C-ABI library loads and exports every symbol include/hh_abi.h declares (no compute calls
without a GPU), and the Python mirror of hh_config matches the C struct."""
import ctypes as C
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_symbols():
    txt = open(os.path.join(ROOT, "include", "hh_abi.h")).read() + open(os.path.join(ROOT, "include", "hh_policy.h")).read()
    return sorted(set(re.findall(r"\b(hh_[a-z_]+)\s*\(", txt)))


def test_library_exports_all_declared_symbols():
    from hhmarl_2d_amd import _lib
    if not os.path.exists(_lib.LIB_PATH):
        import __graft_entry__ as g
        g.build()
    lib = C.CDLL(_lib.LIB_PATH)
    syms = declared_symbols()
    assert len(syms) >= 12
    for s in syms:
        assert hasattr(lib, s), f"libhh_world.so does not export {s}"
    assert set(_lib.EXPORTS) <= set(syms) | {"hh_observe"}


def test_config_struct_layout_matches_header():
    from hhmarl_2d_amd import _lib
    txt = open(os.path.join(ROOT, "include", "hh_abi.h")).read()
    body = re.search(r"typedef struct hh_config \{(.*?)\} hh_config;", txt, re.S).group(1)
    body = re.sub(r"/\*.*?\*/", "", body, flags=re.S)
    fields = re.findall(r"(int32_t|double|uint64_t)\s+([a-z_0-9]+)\s*;", body)
    ctype = {"int32_t": C.c_int32, "double": C.c_double, "uint64_t": C.c_uint64}
    assert [(n, ctype[t]) for t, n in fields] == list(_lib.HHConfig._fields_)
    import oracle_lib
    assert list(oracle_lib.HHConfig._fields_) == list(_lib.HHConfig._fields_)


def test_world_fails_loudly_without_gpu():
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    from hhmarl_2d_amd.world import World, make_config
    with pytest.raises(RuntimeError):
        World(make_config(n_arenas=4, level=3))


def test_net_weights_struct_layout_matches_header():
    from hhmarl_2d_amd import _lib
    txt = open(os.path.join(ROOT, "include", "hh_policy.h")).read()
    body = re.search(r"typedef struct hh_net_weights \{(.*?)\} hh_net_weights;", txt, re.S).group(1)
    body = re.sub(r"/\*.*?\*/", "", body, flags=re.S)
    names = []
    for decl in body.split(";"):
        decl = decl.strip()
        if not decl:
            continue
        for nm in re.findall(r"\*?\s*([a-z_]+)(?:\[\d+\])?\s*(?:,|$)", decl.split(None, 1 if decl.startswith("int32_t") else 2)[-1]):
            names.append(nm)
    assert names == [f[0] for f in _lib.HHNetWeights._fields_], names
    assert C.sizeof(_lib.HHNetWeights) == 8 + 8 * (3 + 3 + 4 + 2 + 2)   # int32 + padding, then 14 pointers
