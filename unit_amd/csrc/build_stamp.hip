// build_stamp.hip -- the content hash of the sources this library was built from (unit_amd/build.py passes it as
// -DUNIT_SOURCE_HASH="..."). unit_amd/_lib.py compares it with the hash of the sources lying next to the library when it loads
// it: a prebuilt .so that travelled to the GPU box beside NEWER sources (mtimes do not survive every checkout / copy) is refused
// instead of silently measured.
#ifndef UNIT_SOURCE_HASH
#define UNIT_SOURCE_HASH "unstamped"
#endif
extern "C" const char* unit_build_hash(void) { return UNIT_SOURCE_HASH; }
