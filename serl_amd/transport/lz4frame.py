"""LZ4 *frame* compression through the system's liblz4 (ctypes) when the `lz4` Python package is absent.

agentlace frames every message as `lz4.frame.compress(pickle.dumps(msg))`.  The `lz4` wheel is not
installed in this image, but the C library it wraps (liblz4.so.1, LZ4F_* API) is, so the bytes on the wire can be real
LZ4 frames (magic 0x184D2204, content size stored like `lz4.frame.compress`'s default `store_size=True`) instead of a
stand-in codec.  `available()` is False when neither is loadable; the transport then falls back to zlib and says so.
"""
from __future__ import annotations

import ctypes as C
import ctypes.util

_lib = None
_tried = False


class _FrameInfo(C.Structure):
    _fields_ = [("blockSizeID", C.c_int), ("blockMode", C.c_int), ("contentChecksumFlag", C.c_int), ("frameType", C.c_int),
                ("contentSize", C.c_ulonglong), ("dictID", C.c_uint), ("blockChecksumFlag", C.c_int)]


class _Prefs(C.Structure):
    _fields_ = [("frameInfo", _FrameInfo), ("compressionLevel", C.c_int), ("autoFlush", C.c_uint),
                ("favorDecSpeed", C.c_uint), ("reserved", C.c_uint * 3)]


def _load():
    global _lib, _tried
    if _tried:
        return _lib
    _tried = True
    for name in (ctypes.util.find_library("lz4"), "liblz4.so.1", "liblz4.so"):
        if not name:
            continue
        try:
            lib = C.CDLL(name)
            lib.LZ4F_compressFrameBound.restype = C.c_size_t
            lib.LZ4F_compressFrameBound.argtypes = [C.c_size_t, C.c_void_p]
            lib.LZ4F_compressFrame.restype = C.c_size_t
            lib.LZ4F_compressFrame.argtypes = [C.c_void_p, C.c_size_t, C.c_void_p, C.c_size_t, C.c_void_p]
            lib.LZ4F_isError.restype = C.c_uint
            lib.LZ4F_isError.argtypes = [C.c_size_t]
            lib.LZ4F_createDecompressionContext.restype = C.c_size_t
            lib.LZ4F_createDecompressionContext.argtypes = [C.POINTER(C.c_void_p), C.c_uint]
            lib.LZ4F_freeDecompressionContext.restype = C.c_size_t
            lib.LZ4F_freeDecompressionContext.argtypes = [C.c_void_p]
            lib.LZ4F_decompress.restype = C.c_size_t
            lib.LZ4F_decompress.argtypes = [C.c_void_p, C.c_void_p, C.POINTER(C.c_size_t), C.c_void_p, C.POINTER(C.c_size_t), C.c_void_p]
            _lib = lib
            break
        except (OSError, AttributeError):
            continue
    return _lib


def available() -> bool:
    return _load() is not None


MAGIC = b"\x04\x22\x4d\x18"


def compress(data: bytes) -> bytes:
    lib = _load()
    if lib is None:
        raise RuntimeError("liblz4 is not loadable")
    prefs = _Prefs()
    prefs.frameInfo.contentSize = len(data)      # lz4.frame.compress(store_size=True)
    cap = lib.LZ4F_compressFrameBound(len(data), C.byref(prefs))
    dst = C.create_string_buffer(cap)
    n = lib.LZ4F_compressFrame(dst, cap, data, len(data), C.byref(prefs))
    if lib.LZ4F_isError(n):
        raise RuntimeError("LZ4F_compressFrame failed")
    return dst.raw[:n]


def decompress(frame: bytes) -> bytes:
    lib = _load()
    if lib is None:
        raise RuntimeError("liblz4 is not loadable")
    ctx = C.c_void_p()
    if lib.LZ4F_isError(lib.LZ4F_createDecompressionContext(C.byref(ctx), 100)):
        raise RuntimeError("LZ4F_createDecompressionContext failed")
    try:
        src = C.create_string_buffer(frame, len(frame))
        out, pos, rc = [], 0, 1
        chunk = C.create_string_buffer(1 << 20)
        while pos < len(frame):
            dn, sn = C.c_size_t(len(chunk)), C.c_size_t(len(frame) - pos)
            rc = lib.LZ4F_decompress(ctx, chunk, C.byref(dn), C.byref(src, pos), C.byref(sn), None)
            if lib.LZ4F_isError(rc):
                raise ValueError("corrupt LZ4 frame")
            out.append(C.string_at(chunk, dn.value))
            pos += sn.value
            if rc == 0 and sn.value == 0 and dn.value == 0:
                break
        if rc != 0:      # the decoder still expects input: the frame ends early
            raise ValueError("truncated LZ4 frame")
        return b"".join(out)
    finally:
        lib.LZ4F_freeDecompressionContext(ctx)
