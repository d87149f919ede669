"""Build staleness by CONTENT (unit_amd/build.py): every object carries the sha256 of (flags, its source, every header), the library the hash of
everything, stamped into it (unit_build_hash, csrc/build_stamp.hip); the loader refuses a library whose stamp differs from the sources next to it.
(Round 3 decided by mtime: a checkout that rewrites mtimes could pair new sources with an old prebuilt .so on the GPU box.)"""
import os

import pytest

from unit_amd import _lib, build


def test_source_hash_is_a_function_of_flags_and_sources(monkeypatch):
    h = build.source_hash()
    assert len(h) == 64 and int(h, 16) >= 0 and h == build.source_hash()
    monkeypatch.setattr(build, "FLAGS", build.FLAGS + ["-DSOMETHING=1"])
    assert build.source_hash() != h


def test_the_loaded_library_carries_the_hash_of_the_sources_next_to_it():
    build.build()
    assert _lib.lib().unit_build_hash().decode() == build.source_hash()
    assert _lib.build_hash() == build.source_hash()[:16]
    assert open(build.LIB + ".hash").read().strip() == build.source_hash()


def test_a_library_built_from_other_sources_is_refused(monkeypatch):
    build.build()
    monkeypatch.setattr(_lib, "_lib", None)
    monkeypatch.delenv("UNIT_HIP_LIB", raising=False)
    monkeypatch.setattr(build, "source_hash", lambda: "0123456789abcdef" * 4)
    with pytest.raises(_lib.UnitLibError, match="built from other sources"):
        _lib.lib()
    monkeypatch.undo()
    monkeypatch.setattr(_lib, "_lib", None)
    assert _lib.lib() is not None          # and loads again once the sources match


def test_build_is_a_no_op_when_nothing_changed():
    build.build()
    t = os.path.getmtime(build.LIB)
    build.build()
    assert os.path.getmtime(build.LIB) == t
