/* hostfault.c — the countdown behind hostfault.h (tests only): freesasa_host_test_fail_after(n) arms it, every
 * allocation / thread creation of the library's host code asks freesasa_hostfault_hit().
 * ref: tests/tools.c:10-48 (the reference's interposed malloc with `fail_after`). */
#include "hostfault.h"

static int g_host_fail_after = 0; /* 0: not armed; k > 0: the k-th call from now on fails */

/* Arms the hook (n <= 0: disarms it).  Returns what was left of the previous countdown: 0 means it fired (or was
 * never armed), k > 0 that the code under test made fewer than that many allocations - the walk 1, 2, ... is over. */
int freesasa_host_test_fail_after(int n)
{
    return __atomic_exchange_n(&g_host_fail_after, n > 0 ? n : 0, __ATOMIC_SEQ_CST);
}

int freesasa_hostfault_hit(void)
{
    int v = __atomic_load_n(&g_host_fail_after, __ATOMIC_RELAXED);
    while (v > 0) {
        if (__atomic_compare_exchange_n(&g_host_fail_after, &v, v - 1, 1, __ATOMIC_SEQ_CST, __ATOMIC_RELAXED)) return v == 1;
    }
    return 0;
}
