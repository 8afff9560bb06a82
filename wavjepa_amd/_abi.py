"""ctypes binding of libwavjepa_hip.so.

The argument structs are generated from `include/wavjepa_hip.h` itself (a tiny parser for its plain-C struct
syntax), so the Python mirror cannot drift from the header; sizes are cross-checked against the library's own
`wj_struct_size`.  There is NO fallback: if the shared library is missing or a symbol is absent, importing the
product path fails loudly.
"""
from __future__ import annotations

import ctypes
import os
import re
from typing import Dict, List, Tuple

HERE = os.path.dirname(os.path.abspath(__file__))
HEADER = os.path.join(os.path.dirname(HERE), "include", "wavjepa_hip.h")
LAB_HEADER = os.path.join(os.path.dirname(HERE), "include", "wavjepa_hip_lab.h")
LIB_PATH = os.environ.get("WAVJEPA_HIP_LIB") or os.path.join(HERE, "lib", "libwavjepa_hip.so")   # override: A/B and ablation builds
# the laboratory build of the same sources (-DWJ_LAB): extra diagnostic entries + every A/B switch; loaded only by load_lab()
LAB_LIB_PATH = os.environ.get("WAVJEPA_HIP_LAB_LIB") or os.path.join(HERE, "lib", "libwavjepa_hip_lab.so")

_SCALARS = {"int64_t": ctypes.c_int64, "int32_t": ctypes.c_int32, "float": ctypes.c_float, "int": ctypes.c_int}


class WavJepaHipError(RuntimeError):
    pass


def _strip_comments(text: str) -> str:
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return re.sub(r"//[^\n]*", "", text)


def parse_header(path: str = HEADER) -> Tuple[Dict[str, List[Tuple[str, object]]], List[str], Dict[str, int]]:
    """Returns ({struct name: [(field, ctype)]}, [function names], {enum constant: value})."""
    text = _strip_comments(open(path).read())
    defines = {k: int(v, 0) for k, v in re.findall(r"^\s*#\s*define\s+(WJ_\w+)\s+(-?(?:0x[0-9a-fA-F]+|\d+))\s*$", text, flags=re.M)}
    structs: Dict[str, List[Tuple[str, object]]] = {}
    for body, name in re.findall(r"typedef\s+struct\s*\{(.*?)\}\s*(\w+)\s*;", text, flags=re.S):
        fields: List[Tuple[str, object]] = []
        for decl in body.split(";"):
            decl = decl.strip()
            if not decl:
                continue
            is_ptr = "*" in decl
            toks = decl.replace("*", " * ").split()
            toks = [t for t in toks if t != "const"]
            base = toks[0]
            names = "".join(toks[1:]).replace("*", "").split(",")
            for n in names:
                n = n.strip()
                if not n:
                    continue
                ctype = ctypes.c_void_p if is_ptr else _SCALARS[base]
                arr = re.match(r"(\w+)\[(\w+)\]$", n)          # fixed-size array member: name[N] (N a number or a #define)
                if arr:
                    n, dim = arr.group(1), arr.group(2)
                    ctype = ctype * (int(dim) if dim.isdigit() else defines[dim])
                fields.append((n, ctype))
        structs[name] = fields
    funcs = re.findall(r"\bint\s+(wj_\w+)\s*\(", text)
    enums: Dict[str, int] = {}
    for body in re.findall(r"enum\s*\{(.*?)\}\s*;", text, flags=re.S):
        val = -1
        for item in body.split(","):
            item = item.strip()
            if not item:
                continue
            if "=" in item:
                k, v = item.split("=")
                val = int(v.strip(), 0)
                enums[k.strip()] = val
            else:
                val += 1
                enums[item] = val
    return structs, funcs, enums


def parse_defines(path: str = HEADER) -> Dict[str, int]:
    """Integer `#define NAME value` constants of the header (WJ_ABI_VERSION, WJ_GROUP_STATS_SPLIT, ...)."""
    text = _strip_comments(open(path).read())
    return {k: int(v, 0) for k, v in re.findall(r"^\s*#\s*define\s+(WJ_\w+)\s+(-?(?:0x[0-9a-fA-F]+|\d+))\s*$", text, flags=re.M)}


_STRUCT_FIELDS, FUNCTIONS, ENUMS = parse_header()
DEFINES = parse_defines()
_LAB_STRUCT_FIELDS, LAB_FUNCTIONS, _ = parse_header(LAB_HEADER)


def _make_struct(name: str, fields):
    return type(name, (ctypes.Structure,), {"_fields_": fields})


STRUCTS = {name: _make_struct(name, fields) for name, fields in _STRUCT_FIELDS.items()}
LAB_STRUCTS = {name: _make_struct(name, fields) for name, fields in _LAB_STRUCT_FIELDS.items()}
_NO_STREAM_FUNCS = {"wj_abi_version": [], "wj_device_count": [], "wj_struct_size": [ctypes.c_char_p],
                    "wj_debug_persist_stamps": [ctypes.c_void_p, ctypes.c_int], "wj_gemm_release_stream": [ctypes.c_void_p],
                    "wj_ln_bwd_partial_rows": [ctypes.c_int, ctypes.c_int], "wj_scatter_fill_bwd_partial_rows": [ctypes.c_int, ctypes.c_int],
                    "wj_rccl_unique_id": [ctypes.c_void_p],
                    "wj_rccl_bucket_allreduce_init": [ctypes.c_void_p], "wj_rccl_bucket_allreduce_finalize": []}

_lib = None
_lab_lib = None


def lib_path() -> str:
    return LIB_PATH


def load_lab():
    """The laboratory library (include/wavjepa_hip_lab.h): the release entries plus the diagnostic ones.  Used by tools/, the all-reduce
    rehearsal (bench.py --emulate-allreduce) and the tests that dissect a kernel; never by the training / inference path."""
    global _lab_lib
    if _lab_lib is None:
        _lab_lib = _load(LAB_LIB_PATH, FUNCTIONS + LAB_FUNCTIONS, {**STRUCTS, **LAB_STRUCTS})
    return _lab_lib


def load():
    """Load the library (once).  Raises WavJepaHipError when it is not built."""
    global _lib
    if _lib is None:
        _lib = _load(LIB_PATH, FUNCTIONS, STRUCTS)
    return _lib


def _load(LIB_PATH: str, FUNCTIONS, STRUCTS):
    if not os.path.exists(LIB_PATH):
        raise WavJepaHipError(
            f"{LIB_PATH} is missing: the HIP extension is not built. Run `python -m wavjepa_amd.build` "
            "(or __graft_entry__.build()). wavjepa_amd has no CPU/PyTorch fallback by design.")
    # PyTorch-ROCm ships its own libamdhip64; the library must bind to THAT runtime (the one that owns the tensors and streams
    # it is handed), so torch is loaded first and the dynamic linker resolves our libamdhip64 dependency to the copy already
    # in the process.  Loaded the other way round there are two HIP runtimes and the first launch fails with "no ROCm-capable
    # device is detected".
    import torch  # noqa: F401
    try:
        lib = ctypes.CDLL(LIB_PATH)
    except OSError as e:  # pragma: no cover
        raise WavJepaHipError(f"cannot load {LIB_PATH}: {e}") from e
    for fn in FUNCTIONS:
        if not hasattr(lib, fn):
            raise WavJepaHipError(f"{LIB_PATH} does not export {fn} (declared in include/wavjepa_hip.h); rebuild it")
        f = getattr(lib, fn)
        f.restype = ctypes.c_int
        if fn in _NO_STREAM_FUNCS:
            f.argtypes = _NO_STREAM_FUNCS[fn]
        else:
            f.argtypes = [ctypes.c_void_p, ctypes.c_void_p]
    if lib.wj_abi_version() != DEFINES["WJ_ABI_VERSION"]:
        raise WavJepaHipError(f"{LIB_PATH} reports ABI version {lib.wj_abi_version()}, include/wavjepa_hip.h declares "
                              f"{DEFINES['WJ_ABI_VERSION']}: stale library, rebuild it (python -m wavjepa_amd.build)")
    if not hasattr(lib, "wj_workspace_bytes"):
        raise WavJepaHipError(f"{LIB_PATH} does not export wj_workspace_bytes; rebuild it")
    lib.wj_workspace_bytes.restype = ctypes.c_int64
    lib.wj_workspace_bytes.argtypes = [ctypes.c_char_p, ctypes.c_void_p]
    for name, cls in STRUCTS.items():
        want = lib.wj_struct_size(name.encode())
        if want != ctypes.sizeof(cls):
            raise WavJepaHipError(f"struct {name}: header mirror is {ctypes.sizeof(cls)} bytes, library says {want}")
    return lib


def workspace_bytes(fn_name: str, args_struct) -> int:
    """Scratch bytes the call `fn_name(args_struct)` needs (pointers in the struct are ignored)."""
    n = int(load().wj_workspace_bytes(fn_name.encode(), ctypes.byref(args_struct)))
    if n < 0:
        raise WavJepaHipError(f"wj_workspace_bytes: unknown entry point {fn_name}")
    return n


_ERR = {-1: "invalid argument", -2: "kernel launch failed", -3: "unsupported configuration"}


def call(fn_name: str, args_struct, stream: int, lab: bool = False) -> None:
    rc = getattr(load_lab() if lab else load(), fn_name)(ctypes.byref(args_struct), stream)
    if rc != 0:
        raise WavJepaHipError(f"{fn_name} failed: {_ERR.get(rc, rc)}")
