#!/usr/bin/env python3
"""Mint the selection-language vectors: freesasa_select_area() of the REFERENCE LIBRARY
(oracle/_ref/libfreesasa_ref.so) for a list of commands — the ones of the reference's own
tests/test_selection.c plus corner cases — on a few structures.  The per-atom "SASA" handed to the
reference is a seeded random weight vector, so that the returned area identifies the selected atom
set (the test recomputes the same weights and sums its own mask).  Build container only.
Output: tests/golden/select.json."""
import ctypes as C
import json
import os

import numpy as np

import make_ingest_golden as mg      # reference library handles, fixture paths

HERE = os.path.dirname(os.path.abspath(__file__))
lib, libc = mg.lib, mg.libc


class Params(C.Structure):
    _fields_ = [("alg", C.c_int), ("probe_radius", C.c_double), ("sr_n", C.c_int), ("lr_n", C.c_int), ("n_threads", C.c_int)]


class Result(C.Structure):            # ref: src/freesasa.h:267-272
    _fields_ = [("total", C.c_double), ("sasa", C.POINTER(C.c_double)), ("n_atoms", C.c_int), ("parameters", Params)]


lib.freesasa_select_area.restype = C.c_int
lib.freesasa_select_area.argtypes = [C.c_char_p, C.c_char_p, C.POINTER(C.c_double), C.c_void_p, C.POINTER(Result)]

COMMANDS = [
    # tests/test_selection.c
    "c1, name ca+o", "c2, name ca", "c3, name oxt", "c4, name ca AND name o", "c5, name ca OR  name o", "c6, name ca+o+oxt",
    "c7, name o5'", "c8, name o5'+ca", "c1, name ABCDE",
    "c1, symbol o+c", "c2, symbol O", "c3, symbol C", "c4, symbol O AND symbol C", "c5, symbol O OR symbol C", "c6, symbol O+C+SE",
    "c7, symbol SE", "c8, symbol O+C+SE and not symbol se", "c1, symbol ABC", "c1, symbol 1", "c1, symbol &%",
    "c1, resn ala+arg", "c2, resn ala", "c3, resn arg", "c4, resn ala AND resn arg", "c5, resn ala OR  resn arg",
    "c6, resn ala+arg AND NOT resn arg", "c1, resn ABCD",
    "c1, resi \\-2-5", "c2, resi 2-4", "c3, resi 1", "c4, resi \\-2 AND resi \\-1-5", "c5, resi \\-2 OR  resi \\-1-5", "c6, resi \\-2-2+2-5",
    "c7, resi \\-1+\\-2-4+5", "c8, resi \\-2-2+7+9+3-5+100", "c9, resi 1-4 AND NOT resi 2-4", "c10,resi \\-2-", "c11,resi -5",
    "c12,resi \\-2-2+2-5", "c13,resi -5 AND NOT resi \\-2+\\-1+1+5", "c14,resi 1-2+3- AND NOT resi 5", "c15,resi 2- AND NOT resi 5",
    "c1, resi 1A", "c1, resi 1a", "c1, resi A", "c1, resi A1", "c1, resi 1AA", "c1, resi 1aa", "c1, resi 1-A", "c1, resi 1A-2",
    "c1, chain A+B", "c2, chain A", "c3, chain B", "c4, chain A AND chain B", "c5, chain A OR chain B", "c6, chain A-B",
    "c7, chain A-B AND NOT chain A", "c1, chain AA", "c1, chain A-1", "c1, chain &",
    "", "a", "a,", "a,b", "a,resi", "a,resn", "a,name", "a,symbol", "a,chain", ",resn ala", ",resi 1", ",name ca", ", symbol c", ",chain a",
    "resn ala", "resi 1", "name ca", "symbol c", "chain a", "resn, ala", "resi, 1", "name, ca", "symbol, c", "chain, a",
    "a, resn ala-arg", "a, name ca-cb", "a, symbol c-o", "a, resi 1-2-3", "a, resi -1-2", "a, resi 1-2-", "a, chain A-",
    "a, resn ala+", "a, resn ala+arg+",
    "c, name ca", "1, name ca", "c1, name ca", "1c, name ca", "-1, name ca", "-1+2_abc, name ca",
    # more of this project's making
    "s, (resn ala or resn gly) and not (name n+c+o)", "s, not resn ala and chain A", "s, not (resn ala and chain A)",
    "s, resn ala or resn gly and name ca", "s, (resn ala or resn gly) and name ca", "s, NAME CA & RESN lys | resn GLU", "s, !symbol c",
    "s, resi 10-20+30+40-45 and symbol n+o", "s, resi 70-", "s, resi -3", "s, chain 1-5", "s, chain 1", "s, resn hoh", "s, symbol fe+zn+se",
    "s, name ca and resi 1-76 and chain A and symbol c and resn met+gln", "s, resname ala", "s, resn ala android", "s, resn a1a",
    "a_very_long_selection_name_that_is_longer_than_fifty_characters_in_total, name ca", "s , name ca", "s,name   ca  ", "s, name ca )",
    "s, ((name ca))", "s, name ca and", "s, and name ca", "s, name 123", "s, resi 1 2", "s, resi 1+2+3-5+\\-1", "s, chain a", "s, chain A+b",
    "s, name c1'+c2'", "s, resn da+dc+a", "s, symbol p", "s, resi 1a+1", "s, resi 52A-53", "s, name ca or name cb or name cg and not resn phe",
]
STRUCTURES = [("1ubq.pdb", 0), ("1ubq.pdb", 1), ("icode.pdb", 0), ("alt_model_twochain.pdb", 0), ("syn_unknowns.pdb", 1),
              ("syn_altloc_icode_chain.pdb", 32), ("3bkr.cif", 1)]


def main():
    out = []
    for name, opt in STRUCTURES:
        path = os.path.join(HERE, "cif" if name.endswith(".cif") else "pdb", name)
        fp = libc.fopen(path.encode(), b"r")
        s = mg.from_cif(fp, None, opt) if name.endswith(".cif") else lib.freesasa_structure_from_pdb(fp, None, opt)
        libc.fclose(fp)
        n = lib.freesasa_structure_n(s)
        seed = 1000 + len(out)
        w = np.random.default_rng(seed).uniform(0.0, 100.0, n)
        res = Result()
        res.sasa = w.ctypes.data_as(C.POINTER(C.c_double))
        res.n_atoms = n
        rows = []
        for cmd in COMMANDS:
            nm = C.create_string_buffer(64)
            area = C.c_double(0)
            rc = lib.freesasa_select_area(cmd.encode(), nm, C.byref(area), s, C.byref(res))
            rows.append({"command": cmd, "rc": rc, "name": nm.value.decode(errors="replace"), "area": area.value.hex()})
        out.append({"file": name, "options": opt, "n_atoms": n, "seed": seed, "selections": rows})
        lib.freesasa_structure_free(s)
    with open(os.path.join(HERE, "select.json"), "w") as fh:
        json.dump(out, fh, indent=1)
    print(sum(len(o["selections"]) for o in out), "vectors;", sum(r["rc"] == -1 for o in out for r in o["selections"]), "failures,",
          sum(r["rc"] == -2 for o in out for r in o["selections"]), "warnings")


if __name__ == "__main__":
    main()
