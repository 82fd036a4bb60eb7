/* hostfault_new.cpp — operator new / delete of libfreesasa_amd.so's OWN C++ code (the engine's std::vector, std::string,
 * std::thread state ...): plain malloc / free behind the fault-injection countdown of hostfault.h, so that the tests can
 * make the n-th allocation of a driver throw std::bad_alloc and watch it come back as an error code (the boundary
 * contract: NULL / FREESASA_FAIL with a message, never an exception, never exit(); ref: src/util.c:89-113).
 * exports.map keeps these symbols LOCAL to the shared library: a host program's operator new is not replaced, and the
 * static seam archive (libfreesasa_amd_seam.a, linked into the reference's own build) does not contain this file. */
#include <new>
#include <stdlib.h>

#include "hostfault.h"

static void *hf_new(size_t n)
{
    if (!freesasa_hostfault_hit()) {
        if (void *p = malloc(n ? n : 1)) return p;
    }
    throw std::bad_alloc();
}
void *operator new(size_t n) { return hf_new(n); }
void *operator new[](size_t n) { return hf_new(n); }
void *operator new(size_t n, const std::nothrow_t &) noexcept { return freesasa_hostfault_hit() ? nullptr : malloc(n ? n : 1); }
void *operator new[](size_t n, const std::nothrow_t &) noexcept { return freesasa_hostfault_hit() ? nullptr : malloc(n ? n : 1); }
void operator delete(void *p) noexcept { free(p); }
void operator delete[](void *p) noexcept { free(p); }
void operator delete(void *p, size_t) noexcept { free(p); }
void operator delete[](void *p, size_t) noexcept { free(p); }
void operator delete(void *p, const std::nothrow_t &) noexcept { free(p); }
void operator delete[](void *p, const std::nothrow_t &) noexcept { free(p); }
