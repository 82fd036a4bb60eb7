/* gpu_parse.h — what gpu_parse.hip (the device-side PDB / mmCIF parser) shares with the sweep driver. */
#ifndef FREESASA_AMD_GPU_PARSE_H
#define FREESASA_AMD_GPU_PARSE_H

#include <stddef.h>

struct freesasa_gpu_ctx;

enum { PARSE_PDB = 0, PARSE_CIF = 1, PARSE_HOST = 2 };

/* one file of a batch's text */
struct ParseFile {
    unsigned beg;          /* its first byte in the batch's text (files[F].beg = the text's length) */
    unsigned row0;         /* mmCIF: first byte (in the batch's text) of the first row of its _atom_site loop */
    short kind;            /* PARSE_PDB / PARSE_CIF / PARSE_HOST (the host parser reads it: nothing of it is looked at) */
    short ncol;            /* mmCIF: columns of the loop */
    signed char slot[12];  /* mmCIF: column of group_PDB, auth_asym_id, auth_seq_id, pdbx_PDB_ins_code, auth_comp_id, auth_atom_id,
                              label_alt_id, type_symbol, Cartn_x, Cartn_y, Cartn_z, pdbx_PDB_model_num (ingest.c cif_cols) */
    unsigned char no_final_nl; /* PDB: the file's last line had no newline (one was added behind it) */
    unsigned char pad[3];
};

/* Phase 1: text -> per-file atoms / status / refused (host arrays [F]) and the total of kept atoms.  Phase 2: the kept atoms
 * into c->h_xyz / c->h_radii / c->h_counts (classes), which are sized for total + extra_atoms first (the caller appends what
 * the host parser read of the refused files behind them).  0 / -1 (message in the context). */
int parse_batch_dev_begin(freesasa_gpu_ctx *c, unsigned char *h_text, size_t T, const ParseFile *files, int F, int options,
                          int *atoms_out, int *status_out, int *host_out, long long *total_atoms_out);
int parse_batch_dev_finish(freesasa_gpu_ctx *c, long long extra_atoms);

#endif
