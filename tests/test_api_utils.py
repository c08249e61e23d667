"""The "extra utils" of the public API (arrays, counted allocations, queue, ring buffer, thread status, log, name tables) through the
C ABI. Expected behaviour is the reference's: include/luminary/{array,host_memory,queue,ringbuffer,thread_status,log,name_strings}.h
and the argument checks / result codes of the matching src/luminary/*.c files."""
import ctypes as C
import threading
import time

import os

import pytest

import luminary_amd

OK, ARG_NULL, INVALID_ARG, MEMORY_LEAK, OOM, API_EXCEPTION = 0, 1, 3, 4, 5, 7
PROPAGATED = 1 << 63


REF_PATH = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "oracle", "_ref", "libluminary_ref_host.so")


@pytest.fixture(params=["ours", "reference"])
def any_lib(request):
    """The same expectations are run against this library and against the reference's own array.c / host_memory.c / queue.c / ringbuffer.c
    (oracle/_ref, built from /root/reference by oracle/build_ref.sh): what the tests expect IS the reference's behaviour."""
    if request.param == "ours":
        return lib()
    if not os.path.exists(REF_PATH):
        pytest.skip("oracle/_ref not built (no reference sources in this checkout)")
    l = C.CDLL(REF_PATH, mode=os.RTLD_LAZY | os.RTLD_LOCAL)
    for name in ("_host_malloc", "_host_realloc", "_host_free", "_array_create", "_array_resize", "_array_push", "_array_copy", "_array_append", "_array_destroy",
                 "array_clear", "array_get_size", "array_get_num_elements", "_array_set_num_elements", "_queue_create", "queue_push", "queue_push_unique", "queue_pop",
                 "queue_pop_blocking", "queue_set_is_blocking", "_queue_destroy", "_ringbuffer_create", "ringbuffer_allocate_entry", "ringbuffer_release_entry",
                 "_ringbuffer_destroy", "thread_status_create", "thread_status_set_worker_name", "thread_status_get_worker_name", "thread_status_start",
                 "thread_status_get_time", "thread_status_get_string", "thread_status_stop", "thread_status_destroy"):
        getattr(l, name).restype = C.c_uint64
    l.is_reference = True
    return l


def lib():
    l = luminary_amd._lib()
    for name in ("_host_malloc", "_host_realloc", "_host_free", "_array_create", "_array_resize", "_array_push", "_array_copy", "_array_append",
                 "_array_destroy", "array_clear", "array_get_size", "array_get_num_elements", "_array_set_num_elements", "_queue_create", "queue_push",
                 "queue_push_unique", "queue_pop", "queue_pop_blocking", "queue_set_is_blocking", "_queue_destroy", "_ringbuffer_create",
                 "ringbuffer_allocate_entry", "ringbuffer_release_entry", "_ringbuffer_destroy", "thread_status_create", "thread_status_set_worker_name",
                 "thread_status_get_worker_name", "thread_status_start", "thread_status_get_time", "thread_status_get_string", "thread_status_stop",
                 "thread_status_destroy", "luminary_ext_host_memory_in_use", "luminary_ext_get_log"):
        getattr(l, name).restype = C.c_uint64
    return l


TAG = (b"buf", b"test", C.c_uint32(1))


def in_use(l):
    if getattr(l, "is_reference", False):
        return 0  # the reference keeps its allocation total private (host_memory.c); the byte counts are checked on ours only
    v = C.c_uint64()
    assert l.luminary_ext_host_memory_in_use(C.byref(v)) == OK
    return v.value


def test_counted_allocations(any_lib):
    l = any_lib
    base = in_use(l)
    grow = 0 if getattr(l, "is_reference", False) else 1
    p = C.c_void_p()
    assert l._host_malloc(C.byref(p), C.c_size_t(1000), *TAG) == OK and p.value and p.value % 16 == 0
    assert in_use(l) == base + 1000 * grow
    C.memset(p, 0xAB, 1000)
    assert l._host_realloc(C.byref(p), C.c_size_t(4000), *TAG) == OK
    assert in_use(l) == base + 4000 * grow and C.string_at(p, 1000) == b"\xab" * 1000
    assert l._host_free(C.byref(p), *TAG) == OK and p.value is None and in_use(l) == base
    assert l._host_free(C.byref(p), *TAG) == ARG_NULL
    assert l._host_malloc(None, C.c_size_t(8), *TAG) == ARG_NULL
    raw = (C.c_uint8 * 256)()
    not_ours = C.c_void_p(C.addressof(raw) + 128)
    assert l._host_free(C.byref(not_ours), *TAG) == API_EXCEPTION


def test_array_growth_append_and_errors(any_lib):
    l = any_lib
    base = in_use(l)
    a = C.c_void_p()
    assert l._array_create(C.byref(a), C.c_size_t(4), C.c_uint32(2), *TAG) == OK
    n, size = C.c_uint32(), C.c_size_t()
    for v in range(5):  # 2 -> 4 -> 8 slots
        x = C.c_uint32(v * 3)
        assert l._array_push(C.byref(a), C.byref(x), *TAG) == OK
    l.array_get_num_elements(a, C.byref(n)); l.array_get_size(a, C.byref(size))
    assert (n.value, size.value) == (5, 8)
    assert list((C.c_uint32 * 5).from_address(a.value)) == [0, 3, 6, 9, 12]
    b = C.c_void_p()
    assert l._array_create(C.byref(b), C.c_size_t(4), C.c_uint32(1), *TAG) == OK
    assert l._array_append(C.byref(b), a, *TAG) == OK and l._array_append(C.byref(b), a, *TAG) == OK
    l.array_get_num_elements(b, C.byref(n))
    assert n.value == 10 and list((C.c_uint32 * 10).from_address(b.value))[5:] == [0, 3, 6, 9, 12]
    assert l._array_copy(C.byref(b), a, *TAG) == OK
    l.array_get_num_elements(b, C.byref(n))
    assert n.value == 5
    assert l._array_set_num_elements(C.byref(b), C.c_uint32(12), *TAG) == OK  # grows and zero-fills
    assert list((C.c_uint32 * 12).from_address(b.value))[5:] == [0] * 7
    assert l._array_resize(C.byref(b), C.c_size_t(3), *TAG) == OK  # shrinking truncates
    l.array_get_num_elements(b, C.byref(n))
    assert n.value == 3
    w = C.c_void_p()
    assert l._array_create(C.byref(w), C.c_size_t(8), C.c_uint32(1), *TAG) == OK
    assert l._array_append(C.byref(w), a, *TAG) == API_EXCEPTION  # element sizes differ
    raw = (C.c_uint8 * 256)()
    assert l.array_clear(C.c_void_p(C.addressof(raw) + 128)) == API_EXCEPTION  # not an array
    assert l.array_clear(a) == OK
    l.array_get_num_elements(a, C.byref(n))
    assert n.value == 0
    for arr in (a, b, w):
        assert l._array_destroy(C.byref(arr), *TAG) == OK and arr.value is None
    assert in_use(l) == base


def test_queue_fifo_unique_and_blocking(any_lib):
    l = any_lib
    q = C.c_void_p()
    assert l._queue_create(C.byref(q), C.c_size_t(0), C.c_size_t(4), *TAG) == INVALID_ARG
    assert l._queue_create(C.byref(q), C.c_size_t(4), C.c_size_t(3), *TAG) == OK
    ok, out = C.c_bool(), C.c_uint32()
    assert l.queue_pop(q, C.byref(out), C.byref(ok)) == OK and not ok.value
    for v in (7, 8, 9):
        x = C.c_uint32(v)
        assert l.queue_push(q, C.byref(x)) == OK
    x = C.c_uint32(10)
    assert l.queue_push(q, C.byref(x)) == OOM  # full
    got = []
    for _ in range(2):
        l.queue_pop(q, C.byref(out), C.byref(ok)); got.append(out.value)
    assert got == [7, 8]
    eq = C.CFUNCTYPE(C.c_bool, C.c_void_p, C.c_void_p)(lambda a, b: C.c_uint32.from_address(a).value == C.c_uint32.from_address(b).value)
    dup = C.c_bool()
    x = C.c_uint32(9)
    assert l.queue_push_unique(q, C.byref(x), eq, C.byref(dup)) == OK and dup.value  # wraps around the ring, already there
    x = C.c_uint32(11)
    assert l.queue_push_unique(q, C.byref(x), eq, C.byref(dup)) == OK and not dup.value
    assert l._queue_destroy(C.byref(q), *TAG) == API_EXCEPTION  # not empty
    l.queue_pop(q, C.byref(out), C.byref(ok)); l.queue_pop(q, C.byref(out), C.byref(ok))
    assert out.value == 11

    # a blocked consumer is released by a push, and by switching blocking off
    res = []

    def consumer():
        o, s = C.c_uint32(), C.c_bool()
        l.queue_pop_blocking(q, C.byref(o), C.byref(s)); res.append((s.value, o.value))
        l.queue_pop_blocking(q, C.byref(o), C.byref(s)); res.append((s.value, None))

    t = threading.Thread(target=consumer)
    t.start()
    time.sleep(0.1)
    x = C.c_uint32(42)
    assert l.queue_push(q, C.byref(x)) == OK
    time.sleep(0.1)
    assert res == [(True, 42)]
    assert l.queue_set_is_blocking(q, C.c_bool(False)) == OK
    t.join(5)
    assert not t.is_alive() and res[1] == (False, None)
    assert l._queue_destroy(C.byref(q), *TAG) == OK and q.value is None


def test_ringbuffer_wraps_without_splitting_entries(any_lib):
    l = any_lib
    r = C.c_void_p()
    assert l._ringbuffer_create(C.byref(r), C.c_size_t(0), *TAG) == INVALID_ARG
    assert l._ringbuffer_create(C.byref(r), C.c_size_t(100), *TAG) == OK
    e = [C.c_void_p() for _ in range(4)]
    assert l.ringbuffer_allocate_entry(r, C.c_size_t(40), C.byref(e[0])) == OK
    assert l.ringbuffer_allocate_entry(r, C.c_size_t(40), C.byref(e[1])) == OK and e[1].value == e[0].value + 40
    assert l.ringbuffer_allocate_entry(r, C.c_size_t(40), C.byref(e[2])) == OOM  # 120 > 100
    assert l.ringbuffer_release_entry(r, C.c_size_t(40)) == OK
    # wraps to the start: the skipped 20-byte tail + 40 live + 40 new = 100 still fits
    assert l.ringbuffer_allocate_entry(r, C.c_size_t(40), C.byref(e[2])) == OK and e[2].value == e[0].value
    assert l.ringbuffer_allocate_entry(r, C.c_size_t(30), C.byref(e[3])) == OOM  # 80 live + 30
    assert l.ringbuffer_release_entry(r, C.c_size_t(90)) == INVALID_ARG  # more than is live
    assert l._ringbuffer_destroy(C.byref(r), *TAG) == MEMORY_LEAK
    assert l.ringbuffer_release_entry(r, C.c_size_t(40)) == OK and l.ringbuffer_release_entry(r, C.c_size_t(40)) == OK
    # fragmentation: 30 bytes live at [40, 70); 70 more fit in total but not contiguously (30-byte tail skipped)
    assert l.ringbuffer_allocate_entry(r, C.c_size_t(30), C.byref(e[3])) == OK and e[3].value == e[0].value + 40
    assert l.ringbuffer_allocate_entry(r, C.c_size_t(70), C.byref(e[2])) == OOM
    assert l.ringbuffer_release_entry(r, C.c_size_t(30)) == OK
    assert l._ringbuffer_destroy(C.byref(r), *TAG) == OK and r.value is None


def test_thread_status_reports_activity(any_lib):
    l = any_lib
    s = C.c_void_p()
    assert l.thread_status_create(C.byref(s)) == OK
    name, text, t = C.c_char_p(), C.c_char_p(), C.c_double()
    keep = C.c_char_p(b"Device 0")
    assert l.thread_status_set_worker_name(s, keep) == OK and l.thread_status_get_worker_name(s, C.byref(name)) == OK and name.value == b"Device 0"
    assert l.thread_status_get_string(s, C.byref(text)) == OK and text.value is None
    work = C.c_char_p(b"Tracing")
    assert l.thread_status_start(s, work) == OK
    x = 0
    t0 = time.process_time()
    while time.process_time() - t0 < 0.05:
        x += 1
    assert l.thread_status_get_time(s, C.byref(t)) == OK and t.value > 0.0
    assert l.thread_status_get_string(s, C.byref(text)) == OK and text.value == b"Tracing"
    assert l.thread_status_stop(s) == OK
    l.thread_status_get_string(s, C.byref(text))
    frozen = C.c_double()
    l.thread_status_get_time(s, C.byref(frozen))
    assert text.value is None and frozen.value >= t.value
    assert l.thread_status_destroy(C.byref(s)) == OK and s.value is None
    assert l.thread_status_destroy(C.byref(s)) == ARG_NULL


def test_log_and_name_tables(tmp_path, monkeypatch):
    l = lib()
    l.luminary_print_log(b"value %d of %s", C.c_int(7), b"seven")
    l.luminary_print_warn(b"careful %0.1f", C.c_double(2.5))
    text, n = C.c_char_p(), C.c_size_t()
    assert l.luminary_ext_get_log(C.byref(text), C.byref(n)) == OK
    assert b"[LOG] value 7 of seven\n" in text.value and b"[WARN] careful 2.5\n" in text.value
    monkeypatch.chdir(tmp_path)
    l.luminary_write_log()
    assert b"value 7 of seven" in (tmp_path / "luminary.log").read_bytes()

    def table(name, count):
        arr = (C.c_char_p * count).in_dll(l, name)
        return [x.decode() for x in arr]

    assert table("luminary_strings_tonemap", 7) == ["None", "ACES", "Reinhard", "Uncharted 2", "Agx", "Agx Punchy", "Agx Custom"]
    assert table("luminary_strings_sky_mode", 3) == ["Default", "HDRI", "Constant Color"]
    assert table("luminary_strings_shading_mode", 6)[5] == "Lights" and table("luminary_strings_filter", 7)[4] == "2 Bit Gray"
    assert table("luminary_strings_material_base_substrate", 2) == ["Opaque", "Translucent"]
    assert table("luminary_strings_aperture", 2) == ["Round", "Bladed"] and len(table("luminary_strings_jerlov_water_type", 10)) == 10
    assert table("luminary_strings_adaptive_sampling_output_mode", 4)[1] == "Rel Variance"


def test_embedded_files_by_name():
    import os
    l = lib()
    l.ceb_access.restype = None
    ptr, n, info = C.c_void_p(), C.c_int64(), C.c_uint64(7)
    l.ceb_access(b"bluenoise_1D.bin", C.byref(ptr), C.byref(n), C.byref(info))
    data_dir = os.path.join(os.path.dirname(luminary_amd.__file__), "data")
    assert info.value == 0 and n.value == 131072 and C.string_at(ptr, n.value) == open(os.path.join(data_dir, "bluenoise_1D.bin"), "rb").read()
    l.ceb_access(b"bluenoise_2D.bin", C.byref(ptr), C.byref(n), C.byref(info))
    assert info.value == 0 and n.value == 262144 and C.string_at(ptr, 64) == open(os.path.join(data_dir, "bluenoise_2D.bin"), "rb").read(64)
    l.ceb_access(b"SplashScreen.bmp", C.byref(ptr), C.byref(n), C.byref(info))
    assert info.value != 0 and ptr.value is None and n.value == 0
