"""ctypes view of libfrhip.so, generated from include/frhip.h at import time.

The header is the single source of truth for the C ABI: struct layouts and prototypes are parsed from it, so
the Python binding cannot drift from what the kernels were compiled against (``self_check`` additionally
compares ``ctypes.sizeof`` with the library's own ``fr_struct_size``).  There is no fallback: if the
shared object is missing the import raises and every op of the product path fails loudly.
"""
import ctypes
import os
import re

import torch  # noqa: F401  -- must precede the CDLL: libfrhip.so has to bind to the HIP runtime torch already loaded
              #                (two runtime copies in one process do not share devices or streams)

_HERE = os.path.dirname(os.path.abspath(__file__))
REPO_ROOT = os.path.dirname(os.path.dirname(_HERE))
HEADER = os.path.join(REPO_ROOT, "include", "frhip.h")
LIB_PATH = os.environ.get("FRHIP_LIB") or os.path.join(_HERE, "lib", "libfrhip.so")  # FRHIP_LIB: A/B builds

_SCALARS = {
    "int": ctypes.c_int, "int32_t": ctypes.c_int32, "int64_t": ctypes.c_int64, "uint64_t": ctypes.c_uint64,
    "uint32_t": ctypes.c_uint32,
    "long long": ctypes.c_longlong, "float": ctypes.c_float, "double": ctypes.c_double,
}


def _ctype(decl, structs):
    """C type text (without the name) -> ctypes type.  Every pointer becomes c_void_p."""
    t = decl.replace("const", " ").strip()
    t = re.sub(r"\s+", " ", t)
    if t.endswith("*"):
        base = t[:-1].strip()
        if base == "char":
            return ctypes.c_char_p
        if base in structs:
            return ctypes.POINTER(structs[base])
        return ctypes.c_void_p
    if t in structs:  # a struct member held by value
        return structs[t]
    return _SCALARS[t]


def parse_header(path=HEADER):
    """Returns (structs: name -> ctypes.Structure subclass, protos: name -> (restype, [argtypes], [argnames]))."""
    src = open(path).read()
    src = re.sub(r"/\*.*?\*/", " ", src, flags=re.S)
    src = re.sub(r"//[^\n]*", " ", src)
    structs = {}
    for m in re.finditer(r"typedef\s+struct\s+(\w+)\s*\{(.*?)\}\s*(\w+)\s*;", src, flags=re.S):
        name, body = m.group(3), m.group(2)
        fields = []
        for stmt in body.split(";"):
            stmt = stmt.strip()
            if not stmt:
                continue
            # "int32_t B, RH, RW"  /  "const float* pro_a"
            mm = re.match(r"(.*?[\s\*])(\w+(?:\s*,\s*\w+)*)$", stmt, flags=re.S)
            ctype = _ctype(mm.group(1), structs)
            for fname in mm.group(2).split(","):
                fields.append((fname.strip(), ctype))
        structs[name] = type(name, (ctypes.Structure,), {"_fields_": fields})
    protos = {}
    body = re.sub(r"typedef\s+struct.*?\}\s*\w+\s*;", " ", src, flags=re.S)
    for m in re.finditer(r"(?:^|;|\})\s*((?:const\s+)?\w+(?:\s+\w+)?\s*\*?)\s*(fr_\w+)\s*\(([^)]*)\)\s*(?=;)", body,
                         flags=re.S):
        ret, name, args = m.group(1).strip(), m.group(2), m.group(3).strip()
        restype = ctypes.c_char_p if "char" in ret else ctypes.c_int
        argtypes, argnames = [], []
        if args and args != "void":
            for a in args.split(","):
                a = a.strip()
                mm = re.match(r"(.*?[\s\*])(\w+)$", a, flags=re.S)
                argtypes.append(_ctype(mm.group(1), structs))
                argnames.append(mm.group(2))
        protos[name] = (restype, argtypes, argnames)
    return structs, protos


structs, protos = parse_header()
FrConvArgs = structs["FrConvArgs"]
FrWgradArgs = structs["FrWgradArgs"]
FrApplyArgs = structs["FrApplyArgs"]
FrBnBwdArgs = structs["FrBnBwdArgs"]
FrSgdTensor = structs["FrSgdTensor"]
FrPackTensor = structs["FrPackTensor"]
FrAdamTensor = structs["FrAdamTensor"]
FrBnEvalEntry = structs["FrBnEvalEntry"]
FrBnFinArgs = structs["FrBnFinArgs"]

# enums of the header
FR_F32, FR_BF16 = 0, 1
PRO_NONE, PRO_BN, PRO_PRELU, PRO_RESBN, PRO_RESBN_SE = 0, 1, 2, 4, 5
EPI_STORE, EPI_STATS, EPI_PRELU_BWD, EPI_BNBWD, EPI_MARGIN, EPI_ATOMIC, EPI_SLAB, EPI_BIAS_RES, EPI_STATS_X = range(9)


class FrhipError(RuntimeError):
    pass


def _load():
    if not os.path.exists(LIB_PATH):
        raise FrhipError(
            "frhip: %s is missing -- build it with `python __graft_entry__.py` (or `make -C %s`). There is no "
            "CPU or PyTorch fallback for the product path." % (LIB_PATH, os.path.join(_HERE, "csrc")))
    lib = ctypes.CDLL(LIB_PATH)
    for name, (restype, argtypes, _names) in protos.items():
        fn = getattr(lib, name)  # AttributeError here == header declares a symbol the library lacks
        fn.restype = restype
        fn.argtypes = argtypes
    return lib


lib = _load()


def self_check():
    assert lib.fr_abi_version() == 7
    for i, s in enumerate((FrConvArgs, FrWgradArgs, FrApplyArgs, FrBnBwdArgs, FrSgdTensor, FrPackTensor, FrAdamTensor, FrBnEvalEntry,
                           FrBnFinArgs)):
        got = lib.fr_struct_size(i)
        if got != ctypes.sizeof(s):
            raise FrhipError("frhip: struct %s is %d bytes in libfrhip.so but %d in the ctypes binding"
                             % (s.__name__, got, ctypes.sizeof(s)))
    return True


def check(rc, what=""):
    if rc != 0:
        msg = lib.fr_last_error_string()
        raise FrhipError("frhip %s failed (rc=%d): %s" % (what, rc, msg.decode() if msg else "?"))
