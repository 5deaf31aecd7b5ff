"""ctypes binding of libpseg_amd.so -- the C-ABI boundary declared in include/pseg_amd.h.

The prototypes are parsed from the header itself, so the binding cannot drift from the ABI.
There is NO fallback: if the library is missing or a call fails this module raises.
"""
import ctypes
import os
import re

_PKG = os.path.dirname(os.path.abspath(__file__))
# (PSEG_LIB_PATH: another build of the same library, e.g. -DPSEG_NO_PRIO=1, for A/B measurements on one box)
LIB_PATH = os.environ.get('PSEG_LIB_PATH') or os.path.join(_PKG, 'libpseg_amd.so')
HEADER_PATH = os.path.join(os.path.dirname(_PKG), 'include', 'pseg_amd.h')

_CTYPES = {
    'int': ctypes.c_int,
    'int64_t': ctypes.c_int64,
    'float': ctypes.c_float,
}


class PsegError(RuntimeError):
    pass


def parse_header(path=HEADER_PATH):
    """-> {name: (restype, [argtypes], [argnames])} for every pseg_* prototype in the header."""
    src = open(path).read()
    src = re.sub(r'/\*.*?\*/', ' ', src, flags=re.S)
    src = re.sub(r'//[^\n]*', ' ', src)
    protos = {}
    for m in re.finditer(r'\b(const\s+char\s*\*|int64_t|int)\s+(pseg_\w+)\s*\(([^)]*)\)\s*;', src):
        ret, name, args = m.group(1), m.group(2), m.group(3).strip()
        restype = ctypes.c_char_p if 'char' in ret else _CTYPES[ret]
        argtypes, argnames = [], []
        if args and args != 'void':
            for a in args.split(','):
                a = ' '.join(a.split())
                am = re.match(r'^(const\s+)?(\w+)\s*(\*?)\s*(\w+)$', a)
                if not am:
                    raise PsegError('cannot parse argument %r of %s' % (a, name))
                base, ptr, an = am.group(2), am.group(3), am.group(4)
                argtypes.append(ctypes.c_void_p if ptr else _CTYPES[base])
                argnames.append(an)
        protos[name] = (restype, argtypes, argnames)
    return protos


def abi_version_of_header(path=HEADER_PATH):
    m = re.search(r'#define\s+PSEG_ABI_VERSION\s+(\d+)', open(path).read())
    if not m:
        raise PsegError('pseg_amd.h does not define PSEG_ABI_VERSION')
    return int(m.group(1))


_lib = None
_protos = None


def load():
    """Load the shared library (once).  Raises PsegError when it is absent -- never falls back."""
    global _lib, _protos
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise PsegError(
            'libpseg_amd.so is missing (%s). Build it with `python -m pytorch_segmentation_amd.csrc.build` '
            '(needs hipcc, cross-compiles for gfx950 without a GPU). There is no CPU or eager fallback.' % LIB_PATH)
    lib = ctypes.CDLL(LIB_PATH)
    _protos = parse_header()
    for name, (restype, argtypes, _) in _protos.items():
        try:
            fn = getattr(lib, name)
        except AttributeError:
            raise PsegError('libpseg_amd.so does not export %s declared in pseg_amd.h (stale build?)' % name)
        fn.restype = restype
        fn.argtypes = argtypes
    want = abi_version_of_header()
    if lib.pseg_abi_version() != want:
        raise PsegError('libpseg_amd.so ABI version %d != %d declared by pseg_amd.h (stale build?)'
                        % (lib.pseg_abi_version(), want))
    _lib = lib
    return lib


def prototypes():
    load()
    return _protos


def call(name, *args):
    """Call an int-returning entry point; raise PsegError with the library's message on failure."""
    lib = load()
    rc = getattr(lib, name)(*args)
    if rc != 0:
        msg = lib.pseg_last_error()
        raise PsegError('%s failed (%d): %s' % (name, rc, msg.decode() if msg else '?'))


_query_cache = {}


def query(name, *args):
    """Call a size/shape query (returns its integer result).  Queries are pure functions of their integer arguments
    (and of the PSEG_* environment): results are memoised, which takes ~3 ctypes round trips off every conv launch."""
    key = (name, args)
    r = _query_cache.get(key)
    if r is None:
        r = _query_cache[key] = getattr(load(), name)(*args)
    return r


def clear_query_cache():
    """After changing a PSEG_CONV_* / PSEG_WGRAD_* planning override at run time: the library re-reads its cached
    environment and the memoised size queries are dropped."""
    _query_cache.clear()
    load().pseg_config_reload()
