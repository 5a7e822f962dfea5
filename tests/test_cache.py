"""pl_cache: the reference's own test scenario (src/tests/cache.c:32-215) replayed through the
C-ABI -- same keys, same sizes, the same expected serialised stream (the reference's golden
bytes for the non-xxhash build: SipHash-2-4 checksums) -- plus stream interchange with the real
reference build where oracle/_ref/libplref.so is present."""
import ctypes as C
import os
import struct

import pytest

import libplacebo_amd as pl

KEY1, KEY2, KEY3, KEY4 = 0x9c65575f419288f5, 0x92da969be9b88086, 0x7fcb62540b00bc8b, 0x46c60ec11af9dde3
KEY5, KEY6, KEY7 = 0xcb6760b98ece2477, 0xf37dc72b7f9e5c88, 0x30c18c962d82e5f5

FREE_FN = C.CFUNCTYPE(None, C.c_void_p)


class Obj(C.Structure):
    _fields_ = [("key", C.c_uint64), ("data", C.c_void_p), ("size", C.c_size_t), ("free", FREE_FN)]




class Params(C.Structure):
    _fields_ = [("log", C.c_void_p), ("max_object_size", C.c_size_t),
                ("max_total_size", C.c_size_t), ("set", C.c_void_p), ("get", C.c_void_p),
                ("priv", C.c_void_p)]


def bind(lib):
    lib.pl_cache_create.restype = C.c_void_p
    lib.pl_cache_create.argtypes = [C.POINTER(Params)]
    lib.pl_cache_destroy.argtypes = [C.POINTER(C.c_void_p)]
    for name, res in (("pl_cache_objects", C.c_int), ("pl_cache_size", C.c_size_t),
                      ("pl_cache_signature", C.c_uint64)):
        getattr(lib, name).restype = res
        getattr(lib, name).argtypes = [C.c_void_p]
    for name in ("pl_cache_try_set", "pl_cache_get"):
        getattr(lib, name).restype = C.c_bool
        getattr(lib, name).argtypes = [C.c_void_p, C.POINTER(Obj)]
    lib.pl_cache_set.restype = None
    lib.pl_cache_set.argtypes = [C.c_void_p, C.POINTER(Obj)]
    lib.pl_cache_save.restype = C.c_size_t
    lib.pl_cache_save.argtypes = [C.c_void_p, C.c_char_p, C.c_size_t]
    lib.pl_cache_load.restype = C.c_int
    lib.pl_cache_load.argtypes = [C.c_void_p, C.c_char_p, C.c_size_t]
    lib.pl_cache_reset.argtypes = [C.c_void_p]
    return lib


@pytest.fixture(scope="module")
def lib(built):
    return bind(pl.lib())


_keep = []


def obj(key, payload):
    buf = C.create_string_buffer(payload, len(payload))
    _keep.append(buf)
    return Obj(key=key, data=C.cast(buf, C.c_void_p), size=len(payload))


def payload(o):
    return C.string_at(o.data, o.size)


def release(o):
    if o.free:
        o.free(o.data)
    o.data, o.size = None, 0


def golden_stream():
    def pad(b):
        return b + b"\0" * (-len(b) % 4)
    s = b"pl_cache" + struct.pack("<II", 1, 2)
    s += struct.pack("<QQQ", KEY3, 4, 0xec18884e5e471117) + pad(b"xyzw")
    s += struct.pack("<QQQ", KEY1, 3, 0x3a204d408a2e2d77) + pad(b"abc")
    return s


def test_reference_scenario(lib):
    test = lib.pl_cache_create(C.byref(Params(max_object_size=16, max_total_size=32)))
    o1, o2, o3 = obj(KEY1, b"abc"), obj(KEY2, b"de"), obj(KEY3, b"xyzw")
    assert lib.pl_cache_signature(test) == 0
    assert lib.pl_cache_try_set(test, o1) and lib.pl_cache_signature(test) == KEY1
    assert lib.pl_cache_try_set(test, o2) and lib.pl_cache_signature(test) == KEY1 ^ KEY2
    assert lib.pl_cache_try_set(test, o3) and lib.pl_cache_signature(test) == KEY1 ^ KEY2 ^ KEY3
    assert (lib.pl_cache_size(test), lib.pl_cache_objects(test)) == (9, 3)
    assert lib.pl_cache_try_set(test, o2)           # ownership moved above: this deletes KEY2
    assert (lib.pl_cache_size(test), lib.pl_cache_objects(test)) == (7, 2)
    assert lib.pl_cache_signature(test) == KEY1 ^ KEY3

    assert lib.pl_cache_get(test, o1) and not lib.pl_cache_get(test, o2) and lib.pl_cache_get(test, o3)
    assert (lib.pl_cache_size(test), lib.pl_cache_objects(test)) == (0, 0)
    assert payload(o1) == b"abc" and payload(o3) == b"xyzw"
    assert lib.pl_cache_try_set(test, o3) and lib.pl_cache_try_set(test, o1)     # reversed order
    assert (lib.pl_cache_size(test), lib.pl_cache_objects(test)) == (7, 2)

    ref = golden_stream()
    assert len(ref) == 72
    data = C.create_string_buffer(100)
    assert lib.pl_cache_save(test, data, 100) == len(ref)
    assert data.raw[:len(ref)] == ref

    test2 = lib.pl_cache_create(C.byref(Params()))
    assert lib.pl_cache_load(test2, data, 100) == 2
    assert lib.pl_cache_signature(test2) == lib.pl_cache_signature(test)
    assert lib.pl_cache_size(test2) == 7
    assert lib.pl_cache_save(test2, None, 0) == len(ref)
    again = C.create_string_buffer(100)
    assert lib.pl_cache_save(test2, again, 100) == len(ref) and again.raw[:len(ref)] == ref
    # invalid streams
    assert lib.pl_cache_load(test2, ref, 0) < 0         # empty
    assert lib.pl_cache_load(test2, ref, 5) < 0         # truncated header
    assert lib.pl_cache_load(test2, ref, 64) == 1       # truncated object data
    bad = bytearray(ref); bad[-2] = ord("X")
    assert lib.pl_cache_load(test2, bytes(bad), len(bad)) == 1     # bad checksum
    t2 = C.c_void_p(test2); lib.pl_cache_destroy(C.byref(t2))

    zero = b"\0" * 32
    o4 = obj(KEY4, zero)
    assert not lib.pl_cache_try_set(test, o4)           # above max_object_size
    assert not lib.pl_cache_get(test, o4)
    assert (lib.pl_cache_size(test), lib.pl_cache_objects(test)) == (7, 2)
    o4 = obj(KEY4, zero[:16])
    assert lib.pl_cache_try_set(test, o4)               # fits, evicts nothing
    assert (lib.pl_cache_size(test), lib.pl_cache_objects(test)) == (23, 3)
    assert lib.pl_cache_get(test, o1) and lib.pl_cache_get(test, o3) and lib.pl_cache_get(test, o4)
    lib.pl_cache_set(test, o1); lib.pl_cache_set(test, o3); lib.pl_cache_set(test, o4)
    assert (lib.pl_cache_size(test), lib.pl_cache_objects(test)) == (23, 3)

    o5 = obj(KEY5, zero[:10])
    assert lib.pl_cache_try_set(test, o5)               # evicts the oldest (KEY1)
    assert (lib.pl_cache_size(test), lib.pl_cache_objects(test)) == (30, 3)
    assert not lib.pl_cache_get(test, o1)
    assert lib.pl_cache_get(test, o3) and lib.pl_cache_get(test, o4) and lib.pl_cache_get(test, o5)
    lib.pl_cache_set(test, o3); lib.pl_cache_set(test, o4); lib.pl_cache_set(test, o5)
    assert (lib.pl_cache_size(test), lib.pl_cache_objects(test)) == (30, 3)

    o6 = obj(KEY6, zero[:6])
    assert lib.pl_cache_try_set(test, o6)               # evicts KEY3
    assert (lib.pl_cache_size(test), lib.pl_cache_objects(test)) == (32, 3)
    assert not lib.pl_cache_get(test, o3)
    assert lib.pl_cache_get(test, o4) and lib.pl_cache_get(test, o5) and lib.pl_cache_get(test, o6)
    assert (lib.pl_cache_size(test), lib.pl_cache_objects(test)) == (0, 0)
    for o in (o4, o5, o6):
        release(o)
    t = C.c_void_p(test); lib.pl_cache_destroy(C.byref(t))
    assert not t.value


def test_callbacks(built):
    """`get` serves misses (the key is forced to the one asked for), `set` sees insertions and
    deletions: tests/c/cache_callbacks.c (ctypes callbacks cannot return structs)"""
    import subprocess
    exe = os.path.join(os.path.dirname(__file__), "c", "build", "cache_callbacks")
    assert os.path.exists(exe), "tests/c/build/cache_callbacks missing: run build()"
    r = subprocess.run([exe], capture_output=True, text=True, timeout=60)
    assert r.returncode == 0 and r.stdout.strip() == "ok", r.stdout + r.stderr


def test_stream_interchange_with_the_reference_build(lib):
    """a stream written by one implementation loads in the other, and is re-emitted bit for bit"""
    path = os.path.join(os.path.dirname(__file__), "..", "oracle", "_ref", "libplref.so")
    if not os.path.exists(path):
        pytest.skip("oracle/_ref/libplref.so not built")
    ref = bind(C.CDLL(path))
    import numpy as np
    rng = np.random.default_rng(7)
    blobs = [(int(rng.integers(1, 2 ** 63)), rng.bytes(int(n))) for n in (1, 2, 3, 4, 5, 63, 64, 1000, 70001)]
    streams = {}
    for name, impl in (("ours", lib), ("ref", ref)):
        c = impl.pl_cache_create(C.byref(Params()))
        for k, b in blobs:
            assert impl.pl_cache_try_set(c, obj(k, b))
        n = impl.pl_cache_save(c, None, 0)
        buf = C.create_string_buffer(n)
        assert impl.pl_cache_save(c, buf, n) == n
        streams[name] = buf.raw
        cc = C.c_void_p(c); impl.pl_cache_destroy(C.byref(cc))
    assert streams["ours"] == streams["ref"]
    for impl in (lib, ref):
        c = impl.pl_cache_create(C.byref(Params()))
        assert impl.pl_cache_load(c, streams["ref"], len(streams["ref"])) == len(blobs)
        for k, b in blobs:
            o = Obj(key=k)
            assert impl.pl_cache_get(c, o) and payload(o) == b
            release(o)
        cc = C.c_void_p(c); impl.pl_cache_destroy(C.byref(cc))


def test_object_files_interchange_with_the_reference_build(lib, tmp_path):
    """pl_cache_set_file / pl_cache_get_file (src/cache.c:474-560): one file per object -- header,
    entry, exactly `size` payload bytes (no padding, unlike a stream) -- written by either
    implementation and read by the other, for sizes that are not multiples of four; no
    directory = no file; an existing file is never overwritten; a corrupt one is removed."""
    path = os.path.join(os.path.dirname(__file__), "..", "oracle", "_ref", "libplref.so")
    if not os.path.exists(path):
        pytest.skip("oracle/_ref/libplref.so not built")
    ref = C.CDLL(path)
    for impl in (lib, ref):
        impl.pl_cache_set_file.restype = None
        impl.pl_cache_set_file.argtypes = [C.c_char_p, Obj]
        impl.pl_cache_get_file.restype = Obj
        impl.pl_cache_get_file.argtypes = [C.c_char_p, C.c_uint64]
    import numpy as np
    rng = np.random.default_rng(11)
    for wname, w, r in (("ours", lib, ref), ("ref", ref, lib)):
        prefix = str(tmp_path / (wname + "_")).encode()
        for n in (1, 2, 3, 5, 7, 64, 1001):
            key, blob = int(rng.integers(1, 2 ** 63)), rng.bytes(n)
            w.pl_cache_set_file(prefix, obj(key, blob))
            name = prefix.decode() + "%016x" % key
            assert os.path.getsize(name) == 16 + 24 + n          # nothing but the payload follows
            for reader in (r, w):
                o = reader.pl_cache_get_file(prefix, key)
                assert o.size == n and payload(o) == blob, (wname, n)
                release(o)
            assert os.path.exists(name)                          # reading never deletes a valid file
    # ours: guards and overwrite policy
    for empty in (None, b""):
        lib.pl_cache_set_file(empty, obj(KEY1, b"abc"))
        assert lib.pl_cache_get_file(empty, KEY1).size == 0
    assert not os.path.exists("%016x" % KEY1)                    # (nothing landed in the CWD)
    prefix = str(tmp_path / "keep_").encode()
    lib.pl_cache_set_file(prefix, obj(KEY2, b"first"))
    lib.pl_cache_set_file(prefix, obj(KEY2, b"second!"))
    o = lib.pl_cache_get_file(prefix, KEY2)
    assert payload(o) == b"first"
    release(o)
    assert not [f for f in os.listdir(tmp_path) if f.endswith(".tmp")]
    name = prefix.decode() + "%016x" % KEY2
    with open(name, "r+b") as f:
        f.seek(16 + 24)
        f.write(b"X")
    assert lib.pl_cache_get_file(prefix, KEY2).size == 0 and not os.path.exists(name)
    lib.pl_cache_set_file(prefix, obj(KEY2, b""))                # size 0 = delete
    assert not os.path.exists(name)
