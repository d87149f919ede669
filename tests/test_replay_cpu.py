"""The generic call behind unit_replay (csrc/replay.hip) and the recorder's argument conversion (unit_amd/_lib.py), without a GPU:
unit_replay_selftest mixes 29 integer-class and 4 float arguments so that the integers overflow onto the stack between floats."""
import ctypes

import pytest

from unit_amd import _lib


def _direct_and_replayed(args):
    l = _lib.lib()
    out_a = (ctypes.c_longlong * 2)()
    out_b = (ctypes.c_longlong * 2)()
    fn = l.unit_replay_selftest
    assert fn(*args, out_a) == 0
    keep = []
    ints, flts = _lib._words(fn, tuple(args) + (out_b,), keep)
    assert len(ints) == 29 and len(flts) == 4 and keep == [out_b]
    arr = (_lib.UnitCall * 1)()
    arr[0].fn, arr[0].n_int, arr[0].n_flt = ctypes.cast(fn, ctypes.c_void_p).value, len(ints), len(flts)
    for k, v in enumerate(ints):
        arr[0].i[k] = v
    for k, v in enumerate(flts):
        arr[0].f[k] = v
    failed = ctypes.c_int(7)
    assert ctypes.sizeof(_lib.UnitCall) == l.unit_call_bytes()
    assert l.unit_replay(arr, 1, ctypes.byref(failed)) == 0 and failed.value == -1
    return list(out_a), list(out_b)


def test_generic_call_places_every_argument():
    buf = (ctypes.c_char * 8)()
    args = [3, ctypes.c_void_p(ctypes.addressof(buf)), 0.125, -(1 << 40), 5, 2.5, -7, 8, (1 << 33) + 9, 10, 11, 7.75] + \
           list(range(12, 30)) + [0.5, 31]
    a, b = _direct_and_replayed(args)
    assert a == b
    want = [3, ctypes.addressof(buf), -(1 << 40), 5, -7, 8, (1 << 33) + 9, 10, 11] + list(range(12, 30)) + [31]
    assert a[0] == sum(v * (i + 1) for i, v in enumerate(want))
    assert a[1] == 125 + 10 * 2500 + 100 * 7750 + 1000 * 500


def test_a_failing_call_is_reported_by_index():
    l = _lib.lib()
    fn = l.unit_replay_selftest
    out = (ctypes.c_longlong * 2)()
    good = [0, None, 0.0, 0, 0, 0.0] + [0] * 5 + [0.0] + [0] * 18 + [0.0, 0]
    arr = (_lib.UnitCall * 3)()
    for j, o in enumerate((out, None, out)):          # the second call has a NULL out pointer: UNIT_ERR_ARG
        ints, flts = _lib._words(fn, tuple(good) + (o,), [])
        arr[j].fn, arr[j].n_int, arr[j].n_flt = ctypes.cast(fn, ctypes.c_void_p).value, len(ints), len(flts)
        for k, v in enumerate(ints):
            arr[j].i[k] = v
    failed = ctypes.c_int(-5)
    assert l.unit_replay(arr, 3, ctypes.byref(failed)) == -1 and failed.value == 1
    assert b"null out" in l.unit_last_error()


def test_which_calls_are_recorded():
    names = _lib.parse_header_names()
    assert set(names) == set(_lib.parse_header())
    for n in ("unit_conv2d_fwd", "unit_conv2d_wgrad_group", "unit_stream_wait_stream", "unit_sgd_momentum", "unit_event_record_raw", "unit_nms"):
        assert _lib.enqueues(n), n
    for n in ("unit_conv2d_wgrad_group_plan", "unit_conv2d_wgrad_splits", "unit_sort_workspace_bytes", "unit_build_hash", "unit_replay"):
        assert not _lib.enqueues(n), n
    # every function that takes a stream has exactly the 32 + 8 words of room
    for n, (_, argtypes) in _lib.parse_header().items():
        assert sum(t is not ctypes.c_float for t in argtypes) <= _lib.UnitCall.INTS and sum(t is ctypes.c_float for t in argtypes) <= _lib.UnitCall.FLOATS


def test_byref_and_struct_arguments_are_pinned():
    from unit_amd import ops
    l = _lib.lib()
    sec = ops.ConvSecond()
    keep = []
    fn = l.unit_conv2d_fwd_pair
    args = [0] + [None] * 6 + [0] * 20 + [ctypes.byref(sec), None]
    assert len(args) == len(fn.argtypes)
    ints, _ = _lib._words(fn, args, keep)
    assert keep == [sec] and ints[-2] == ctypes.addressof(sec)
    with pytest.raises(TypeError):
        _lib._words(fn, args[:-2] + ["not a pointer", None], [])


def test_only_the_recording_thread_is_recorded():
    """while a step is being recorded, `_lib.lib()` hands the recording proxy to the recording thread only: a helper thread of the process (a data
    loader, a collective's worker) keeps talking to the library itself, and a paused recorder steps aside for its own thread too"""
    import threading
    real = _lib.lib()
    seen = {}
    with _lib.Recorder() as rec:
        assert _lib.lib() is rec
        t = threading.Thread(target=lambda: seen.setdefault("other", _lib.lib()))
        t.start()
        t.join()
        with rec.paused():
            seen["paused"] = _lib.lib()
        # host-only calls pass through the proxy unrecorded
        assert rec.unit_call_bytes() == real.unit_call_bytes()
    assert seen["other"] is real and seen["paused"] is real and _lib.lib() is real
    plan = rec.finish()
    assert plan.n_calls == 0 and plan.items == []
