"""ctypes binding of libmmdti_hip.so -- the only way the host package reaches the kernels.

The prototypes are parsed from ``include/mmdti_hip.h`` (the single source of truth for the C ABI), so the Python
side cannot drift from the header.  There is NO fallback: if the shared library is missing or a symbol the header
declares is absent, importing this module raises.
"""
import ctypes
import os
import re

# torch must load ITS bundled HIP runtime (torch/lib/libamdhip64.so) before libmmdti_hip.so is dlopen'ed: both have the
# same SONAME, and the first one loaded serves the whole process.  Loading the system copy first leaves torch and the
# kernels on different runtimes ("no ROCm-capable device is detected").
import torch  # noqa: F401  (plumbing: device memory + streams)

_HERE = os.path.dirname(os.path.abspath(__file__))
REPO_ROOT = os.path.dirname(os.path.dirname(_HERE))
HEADER = os.path.join(REPO_ROOT, "include", "mmdti_hip.h")
LIB_PATH = os.environ.get("MMDTI_HIP_LIB", os.path.join(_HERE, "lib", "libmmdti_hip.so"))


class MMDTIError(RuntimeError):
    pass


def _ctype(decl: str):
    d = decl.strip()
    if "*" in d or d.startswith("mmdti_stream_t"):
        return ctypes.c_void_p
    d = re.sub(r"\b[A-Za-z_][A-Za-z_0-9]*$", "", d).strip() if len(d.split()) > 1 else d   # drop the parameter name
    d = d.replace("const", "").strip()
    table = {
        "int": ctypes.c_int, "unsigned int": ctypes.c_uint, "long long": ctypes.c_longlong,
        "unsigned long long": ctypes.c_ulonglong, "float": ctypes.c_float, "void": None,
    }
    if d not in table:
        raise MMDTIError(f"mmdti_hip.h: unsupported parameter type in {decl!r}")
    return table[d]


def parse_header(path: str = HEADER):
    """-> {name: (restype, [argtypes], [argnames])} for every function the header declares."""
    src = open(path).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    src = re.sub(r"//[^\n]*", "", src)
    protos = {}
    for m in re.finditer(r"(const\s+char\s*\*|int)\s+(mmdti_[a-z0-9_]+)\s*\(([^)]*)\)\s*;", src):
        ret, name, args = m.group(1), m.group(2), m.group(3)
        args = " ".join(args.split())
        if args in ("void", ""):
            argtypes, argnames = [], []
        else:
            parts = [a.strip() for a in args.split(",")]
            argtypes = [_ctype(a) for a in parts]
            argnames = [re.findall(r"[A-Za-z_][A-Za-z_0-9]*", a)[-1] for a in parts]
        protos[name] = (ctypes.c_char_p if "char" in ret else ctypes.c_int, argtypes, argnames)
    if not protos:
        raise MMDTIError(f"no prototypes found in {path}")
    return protos


def header_constants(path: str = HEADER):
    return {k: int(v) for k, v in re.findall(r"#define\s+(MMDTI_[A-Z0-9_]+)\s+(\d+)", open(path).read())}


class _Lib:
    def __init__(self):
        if not os.path.exists(LIB_PATH):
            raise MMDTIError(
                f"libmmdti_hip.so not found at {LIB_PATH}: build it with `python __graft_entry__.py` "
                f"(make -C mm-dti_amd/csrc).  There is no CPU fallback for the product path.")
        self._dll = ctypes.CDLL(LIB_PATH)
        self.protos = parse_header()
        self.const = header_constants()
        for name, (restype, argtypes, _) in self.protos.items():
            try:
                fn = getattr(self._dll, name)
            except AttributeError as e:
                raise MMDTIError(f"{LIB_PATH} does not export {name} declared in include/mmdti_hip.h") from e
            fn.restype = restype
            fn.argtypes = argtypes
        self._last_error = self._dll.mmdti_last_error

    def call(self, name, *args):
        rc = getattr(self._dll, name)(*args)
        if rc != 0:
            msg = self._last_error()
            raise MMDTIError(f"{name} failed (code {rc}): {msg.decode() if msg else '?'}")

    def __getattr__(self, name):
        # (reached once per symbol: the checked caller is cached on the instance, so later lookups skip this method)
        if name.startswith("mmdti_"):
            fn = getattr(self._dll, name)
            if self.protos[name][0] is ctypes.c_int and name != "mmdti_abi_version":
                last_error = self._last_error

                def checked(*a, _fn=fn, _name=name):
                    rc = _fn(*a)
                    if rc != 0:
                        msg = last_error()
                        raise MMDTIError(f"{_name} failed (code {rc}): {msg.decode() if msg else '?'}")
                fn = checked
            self.__dict__[name] = fn
            return fn
        raise AttributeError(name)


_lib = None


def lib() -> _Lib:
    global _lib
    if _lib is None:
        _lib = _Lib()
    return _lib
