/*
 * hostfault.h — host-side fault injection for the library's own host code (tests only; never armed in production).
 *
 * The reference tests its error paths by interposing malloc / realloc / strdup for the whole test process and making
 * the n-th call fail (ref: tests/tools.c:10-48, used by tests/test_freesasa.c:475-514, tests/test_nb.c:29-44).  This
 * library is loaded into Python test processes, so it cannot interpose libc for everyone; instead every allocation and
 * thread creation OF THE LIBRARY'S HOST CODE goes through the wrappers below (C sources), through the library-local
 * operator new of hostfault_new.cpp and through ThreadGroup::spawn / Joiner::start (engine_internal.h; C++ sources),
 * and freesasa_host_test_fail_after(n) makes the n-th of them fail: NULL, std::bad_alloc or a thread that does not
 * start.  Device and page-locked allocations have their own hook (freesasa_gpu_test_fail_after, gpu_engine.hip).
 */
#ifndef FREESASA_AMD_HOSTFAULT_H
#define FREESASA_AMD_HOSTFAULT_H

#include <pthread.h>
#include <stdlib.h>
#include <string.h>

#ifdef __cplusplus
extern "C" {
#endif
/* 1: this allocation / thread creation is the one that has to fail (counts down; 0 when the hook is not armed) */
int freesasa_hostfault_hit(void);
#ifdef __cplusplus
}
#endif

static inline void *hf_malloc(size_t n) { return freesasa_hostfault_hit() ? NULL : malloc(n); }
static inline void *hf_calloc(size_t k, size_t n) { return freesasa_hostfault_hit() ? NULL : calloc(k, n); }
static inline void *hf_realloc(void *p, size_t n) { return freesasa_hostfault_hit() ? NULL : realloc(p, n); }
static inline int hf_thread_create(pthread_t *t, const pthread_attr_t *at, void *(*fn)(void *), void *arg)
{
    return freesasa_hostfault_hit() ? 11 /* EAGAIN */ : pthread_create(t, at, fn, arg);
}

#endif
