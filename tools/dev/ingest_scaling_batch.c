/* Host-side scaling of the loader: argv[1] threads, argv[2] copies of every fixture, argv[3] "cif" for
 * the mmCIF fixtures (default: the PDB ones).  Built by tools/gpu_ingest.sh. */
#define _GNU_SOURCE
#include "../../freesasa_amd/csrc/ingest.c"
#include <glob.h>
int main(int c, char **v)
{
    const int nt = atoi(v[1]);
    const int cif = c > 3 && strcmp(v[3], "cif") == 0;
    glob_t g;
    glob(cif ? "tests/golden/cif/[1-9]*.cif" : "tests/golden/pdb/[1-9]*.pdb", 0, 0, &g);
    const int n = (int)g.gl_pathc * (c > 2 ? atoi(v[2]) : 32);
    const char **paths = malloc(sizeof(char *) * n);
    for (int i = 0; i < n; ++i) paths[i] = g.gl_pathv[i % g.gl_pathc];
    for (int rep = 0; rep < 3; ++rep) {
        freesasa_ingest_batch b;
        struct timespec a, bb;
        clock_gettime(CLOCK_MONOTONIC, &a);
        const int rc = freesasa_ingest_pdb_files(paths, n, 0, nt, &b);
        clock_gettime(CLOCK_MONOTONIC, &bb);
        const double dt = (bb.tv_sec - a.tv_sec) + 1e-9 * (bb.tv_nsec - a.tv_nsec);
        printf("%s %d threads: rc %d %.3f s -> %.1f M atoms/s\n", cif ? "mmCIF" : "PDB", nt, rc, dt, b.n_atoms / dt / 1e6);
        freesasa_ingest_free(&b);
    }
    return 0;
}
