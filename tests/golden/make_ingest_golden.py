#!/usr/bin/env python3
"""Mint the ingestion vectors: what the REFERENCE LIBRARY (oracle/_ref/libfreesasa_ref.so, built
from /root/reference by `make -C oracle ref`) holds after freesasa_structure_from_pdb() for the PDB
files of its own test suite, under the option sets the loader supports.  Runs only in the build
container.  Output:
  tests/golden/pdb/*.pdb      copies of the reference's test DATA files (inputs; big NMR ensembles
                              are cut after their second model)
  tests/golden/ingest.json    per (file, options): atom/residue counts and sha256 digests of the
                              coordinate, radius, class and residue-boundary arrays, plus the
                              residue labels' digest; failures recorded as {"fail": true}
"""
import ctypes as C
import hashlib
import json
import os
import shutil

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
DATA = "/root/reference/tests/data"
lib = C.CDLL(os.path.join(ROOT, "oracle", "_ref", "libfreesasa_ref.so"))
libc = C.CDLL(None)
libc.fopen.restype = C.c_void_p
libc.fopen.argtypes = [C.c_char_p, C.c_char_p]
libc.fclose.argtypes = [C.c_void_p]
lib.freesasa_structure_from_pdb.restype = C.c_void_p
lib.freesasa_structure_from_pdb.argtypes = [C.c_void_p, C.c_void_p, C.c_int]
lib.freesasa_structure_free.argtypes = [C.c_void_p]
for f, rt in (("freesasa_structure_n", C.c_int), ("freesasa_structure_n_residues", C.c_int),
              ("freesasa_structure_coord_array", C.POINTER(C.c_double)), ("freesasa_structure_radius", C.POINTER(C.c_double))):
    getattr(lib, f).restype = rt
    getattr(lib, f).argtypes = [C.c_void_p]
lib.freesasa_structure_atom_class.restype = C.c_int
lib.freesasa_structure_atom_class.argtypes = [C.c_void_p, C.c_int]
lib.freesasa_structure_residue_atoms.argtypes = [C.c_void_p, C.c_int, C.POINTER(C.c_int), C.POINTER(C.c_int)]
lib.freesasa_structure_residue_name.restype = C.c_char_p
lib.freesasa_structure_residue_name.argtypes = [C.c_void_p, C.c_int]
lib.freesasa_structure_residue_number.restype = C.c_char_p
lib.freesasa_structure_residue_number.argtypes = [C.c_void_p, C.c_int]
lib.freesasa_structure_residue_chain_lcl.restype = C.c_char_p
lib.freesasa_structure_residue_chain_lcl.argtypes = [C.c_void_p, C.c_int]
lib.freesasa_structure_atom_name.restype = C.c_char_p
lib.freesasa_structure_atom_name.argtypes = [C.c_void_p, C.c_int]
lib.freesasa_atom_is_backbone.argtypes = [C.c_char_p]
lib.freesasa_structure_residue_reference.restype = C.c_void_p
lib.freesasa_structure_residue_reference.argtypes = [C.c_void_p, C.c_int]
lib.freesasa_set_verbosity(2)  # FREESASA_V_SILENT
# the reference's mmCIF reader (C++ over gemmi), built by `make -C oracle ref` next to the C library
libcif = C.CDLL(os.path.join(ROOT, "oracle", "_ref", "libfreesasa_refcif.so"))
from_cif = getattr(libcif, "_Z27freesasa_structure_from_cifP8_IO_FILEPK19freesasa_classifieri")
from_cif.restype = C.c_void_p
from_cif.argtypes = [C.c_void_p, C.c_void_p, C.c_int]


HETATM, HYDROGEN, JOIN, HALT, SKIP, OCC = 1, 1 << 2, 1 << 5, 1 << 6, 1 << 7, 1 << 8
OPTION_SETS = [0, HETATM, HYDROGEN, HETATM | HYDROGEN, JOIN, SKIP, HALT, HETATM | SKIP, OCC, HETATM | HYDROGEN | JOIN]
FILES = ["1ubq.pdb", "1a0q.pdb", "3bkr.pdb", "3bzd_trimmed.pdb", "5dx9.pdb", "1d3z.pdb", "2jo4.pdb", "3gnn.pdb",
         "alt_model_twochain.pdb", "icode.pdb", "1ubq.occ.pdb", "empty.pdb", "empty_model.pdb", "model_mismatch.pdb",
         "reference_bfactors.pdb", "1ubq.B.pdb"]
CUT_AFTER_MODELS = {"1d3z.pdb": 2, "2jo4.pdb": 2}
CIF_FILES = ["1ubq.cif", "3bkr.cif", "5dx9.cif", "7cma-assembly1.cif"]
CIF_OPTION_SETS = [0, HETATM, HYDROGEN, HETATM | HYDROGEN, JOIN, SKIP, HALT, HETATM | SKIP, HETATM | HYDROGEN | JOIN]


def cif_loop(rows, columns=None, block="data_SYN"):
    cols = columns or ["group_PDB", "id", "type_symbol", "label_atom_id", "label_alt_id", "label_comp_id", "label_asym_id",
                       "label_entity_id", "label_seq_id", "pdbx_PDB_ins_code", "Cartn_x", "Cartn_y", "Cartn_z", "occupancy",
                       "B_iso_or_equiv", "pdbx_formal_charge", "auth_seq_id", "auth_comp_id", "auth_asym_id", "auth_atom_id",
                       "pdbx_PDB_model_num"]
    return block + "\n#\nloop_\n" + "".join(f"_atom_site.{c}\n" for c in cols) + "\n".join(rows) + "\n#\n"


def synthetic_cifs():
    """mmCIF inputs of this project's own making; expected outcome = what the reference does."""
    def row(i, sym, name, alt, comp, asym, seq, ins, xyz, model, group="ATOM", auth_name=None, auth_asym=None):
        return (f"{group} {i} {sym} {name} {alt} {comp} {asym} 1 {seq} {ins} {xyz} 1.00 10.00 ? {seq} {comp} "
                f"{auth_asym or asym} {auth_name or name} {model}")
    files = {}
    files["syn_basic.cif"] = cif_loop([
        row(1, "N", "N", ".", "ALA", "A", 1, "?", "1.000 2.000 3.000", 1),
        row(2, "C", "CA", ".", "ALA", "A", 1, "?", "2.000 2.000 3.000", 1),
        row(3, "C", '"C1\'"', ".", "A", "B", 2, "?", "3.000 2.000 3.000", 1),          # quoted name with a prime
        row(4, "H", "HA", ".", "ALA", "A", 1, "?", "4.000 2.000 3.000", 1),
        row(5, "D", "D1", ".", "ALA", "A", 1, "?", "5.000 2.000 3.000", 1),                # deuterium is not "H"
        row(6, "O", "O", ".", "HOH", "A", 101, "?", "6.000 2.000 3.000", 1, group="HETATM"),
        row(7, "FE", "FE", ".", "HEM", "A", 102, "?", "7.0 -2.5e0 +3.", 1, group="HETATM"),
        row(8, "C", "CA", ".", "UNK", "A", 3, "?", "8.000 2.000 3.000", 1),                # unknown residue
        row(9, "Q", "QQ", ".", "UNK", "A", 3, "?", "9.000 2.000 3.000", 1),                # unknown element
        row(10, "C", "CB", ".", "ALA", "A", 1, "?", "10.000 2.000 3.000", 2),              # other model
    ])
    files["syn_altloc_icode_chains.cif"] = cif_loop([
        row(1, "N", "N", ".", "SER", "A", 1, "?", "1.000 2.000 3.000", 1),
        row(2, "C", "CA", "B", "SER", "A", 1, "?", "2.000 2.000 3.000", 1),
        row(3, "C", "CA", "A", "SER", "A", 1, "?", "2.100 2.000 3.000", 1),
        row(4, "C", "CB", "B", "SER", "A", 1, "?", "3.000 2.000 3.000", 1),
        row(5, "N", "N", ".", "GLY", "A", 1, "A", "5.000 2.000 3.000", 1),                 # insertion code
        row(6, "N", "N", ".", "GLY", "AAAA", 1, "A", "6.000 2.000 3.000", 1),              # long chain id (cut to 3)
        row(7, "C", "CA", ".", "GLY", "AAAB", 1, "A", "7.000 2.000 3.000", 1),             # same after the cut
        row(8, "N", "N", ".", "ALANINE", "C", 123456, "?", "8.000 2.000 3.000", 1),        # long names / numbers are cut
        row(9, "C", "CA", ".", "ALA", "C", 123457, "?", "9.000 2.000 3.000", 1),
    ])
    files["syn_models_out_of_order.cif"] = cif_loop([
        row(1, "N", "N", ".", "ALA", "A", 1, "?", "1.000 2.000 3.000", 3),
        row(2, "C", "CA", ".", "ALA", "A", 1, "?", "2.000 2.000 3.000", 2),
        row(3, "C", "C", ".", "ALA", "A", 1, "?", "3.000 2.000 3.000", 3),
        row(4, "O", "O", ".", "ALA", "A", 1, "?", "4.000 2.000 3.000", 2),
    ])
    two = cif_loop([row(1, "N", "N", ".", "ALA", "A", 1, "?", "1.000 2.000 3.000", 1)], block="data_ONE")
    two += "_cell.length_a 10.0\n_struct.title\n;a text field with loop_ and _atom_site.id inside\n;\n"
    two += cif_loop([row(1, "C", "CA", ".", "GLY", "B", 5, "?", "2.000 2.000 3.000", 1)], block="data_TWO")
    files["syn_two_blocks_textfield.cif"] = two
    # columns in another order, upper-case keywords and tags, values spread over lines, comments
    cols = ["pdbx_PDB_model_num", "Cartn_z", "Cartn_y", "Cartn_x", "type_symbol", "label_alt_id", "auth_atom_id", "auth_comp_id",
            "pdbx_PDB_ins_code", "auth_seq_id", "auth_asym_id", "group_PDB"]
    files["syn_reordered_columns.cif"] = ("DATA_X\nLOOP_\n" + "".join(f"_ATOM_SITE.{c}\n" for c in cols) +
                                          "1 3.0 2.0 1.0 N . N ALA ? 1 A ATOM # trailing comment\n1 3.0 2.0\n 2.0 C . 'CA' ALA\n ? 1 A ATOM\n")
    files["syn_missing_column.cif"] = cif_loop([f"ATOM 1 N N . ALA A 1 ? 1.0 2.0 3.0 1"],
                                               columns=["group_PDB", "id", "type_symbol", "auth_atom_id", "label_alt_id", "auth_comp_id",
                                                        "auth_asym_id", "auth_seq_id", "pdbx_PDB_ins_code", "Cartn_x", "Cartn_y", "Cartn_z",
                                                        "occupancy"])          # no pdbx_PDB_model_num
    files["syn_no_atoms.cif"] = "data_EMPTY\n_cell.length_a 10.0\n"
    # the category written as tag-value pairs (a one-atom file): gemmi's block.find() takes it as a one-row table
    pair_cols = ["group_PDB", "id", "type_symbol", "label_atom_id", "label_alt_id", "label_comp_id", "label_asym_id",
                 "label_seq_id", "pdbx_PDB_ins_code", "Cartn_x", "Cartn_y", "Cartn_z", "auth_seq_id", "auth_comp_id", "auth_asym_id",
                 "auth_atom_id", "pdbx_PDB_model_num"]
    pair_vals = ["ATOM", "1", "N", "N", ".", "ALA", "A", "1", "?", "1.500", "-2.250", "3.125", "7", "ALA", "A", "N", "1"]
    files["syn_pair_form.cif"] = "data_PAIR\n_cell.length_a 10.0\n" + "".join(f"_atom_site.{c} {v}\n" for c, v in zip(pair_cols, pair_vals))
    files["syn_pair_form_incomplete.cif"] = "data_PAIR\n" + "".join(f"_atom_site.{c} {v}\n" for c, v in zip(pair_cols[:-1], pair_vals[:-1]))
    # (a block with two _atom_site loops is not an input the reference defines: gemmi refuses duplicate tags with an
    # exception that src/cif.cc does not catch; the reader here takes the first loop that carries group_PDB)
    return files


def sha(a):
    return hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()


def reference_view(path, options):
    """reference_view_unsafe in a forked child: a few inputs make the reference abort (double free
    when RADIUS_FROM_OCCUPANCY meets a line without the occupancy column, src/structure.c:696-699
    then :717-719); those are recorded as {"crash": true} and not used as vectors."""
    r, w = os.pipe()
    pid = os.fork()
    if pid == 0:
        os.close(r)
        try:
            os.write(w, json.dumps(reference_view_unsafe(path, options)).encode())
        finally:
            os._exit(0)
    os.close(w)
    data = b""
    while True:
        chunk = os.read(r, 65536)
        if not chunk:
            break
        data += chunk
    os.close(r)
    _, status = os.waitpid(pid, 0)
    if status != 0 or not data:
        return {"crash": True}
    return json.loads(data)


def reference_view_unsafe(path, options):
    fp = libc.fopen(path.encode(), b"r")
    if path.endswith(".cif"):
        s = from_cif(fp, None, options)
    else:
        s = lib.freesasa_structure_from_pdb(fp, None, options)
    libc.fclose(fp)
    if not s:
        return {"fail": True}
    n, nr = lib.freesasa_structure_n(s), lib.freesasa_structure_n_residues(s)
    if n == 0:      # the mmCIF reader hands back an empty structure where the PDB reader fails
        return {"fail": True}
    xyz = np.ctypeslib.as_array(lib.freesasa_structure_coord_array(s), shape=(3 * n,)).copy()
    rad = np.ctypeslib.as_array(lib.freesasa_structure_radius(s), shape=(n,)).copy()
    cls = np.array([lib.freesasa_structure_atom_class(s, i) for i in range(n)], dtype=np.uint8)
    bb = np.array([lib.freesasa_atom_is_backbone(lib.freesasa_structure_atom_name(s, i)) for i in range(n)], dtype=np.uint8)
    has_ref = np.array([1 if lib.freesasa_structure_residue_reference(s, r) else 0 for r in range(nr)], dtype=np.uint8)
    first, labels = [], []
    a, b = C.c_int(), C.c_int()
    for r in range(nr):
        lib.freesasa_structure_residue_atoms(s, r, C.byref(a), C.byref(b))
        first.append(a.value)
        labels.append(lib.freesasa_structure_residue_name(s, r).decode() + "|" + lib.freesasa_structure_residue_number(s, r).decode()
                      + "|" + lib.freesasa_structure_residue_chain_lcl(s, r).decode())
    lib.freesasa_structure_free(s)
    return {"n_atoms": n, "n_residues": nr, "xyz": sha(xyz), "radii": sha(rad), "classes": sha(cls),
            "res_first": sha(np.array(first + [n], dtype=np.int64)), "labels": hashlib.sha256("\n".join(labels).encode()).hexdigest(),
            "backbone": sha(bb), "has_reference": sha(has_ref), "radius_sum": float(rad.sum()), "polar": int((cls == 1).sum()), "unknown": int((cls == 2).sum())}


def atom_line(serial, name, res, chain, resnum, coords, tail="  1.00  0.00           C  ", alt=" ", rec="ATOM  ", icode=" "):
    """Fixed-column ATOM record; `coords` is the raw 24-character coordinate section."""
    assert len(coords) == 24 and len(name) == 4 and len(res) == 3
    return f"{rec}{serial:5d} {name}{alt}{res} {chain}{resnum:4d}{icode}   {coords}{tail}"


def synthetic_files():
    """Inputs of this project's own making that poke at the reader's corners; the expected
    outcome is still whatever the reference library does with them."""
    L = atom_line
    files = {}
    files["syn_numbers.pdb"] = "\n".join([
        L(1, " N  ", "ALA", "A", 1, "  11.104   6.134  -6.504", tail="  1.00  0.00           N  "),
        L(2, " CA ", "ALA", "A", 1, "-100.123-200.456-300.789"),          # columns run together
        L(3, " C  ", "ALA", "A", 1, "   1e1    2.5E-1  -3.e0 "),          # exponents
        L(4, " O  ", "ALA", "A", 1, "  +1.5     .25      -.5 ", tail="  1.00  0.00           O  "),
        L(5, " CB ", "ALA", "A", 1, "1.23456789012345678 2. 3"),           # more than 15 digits
        L(6, " N  ", "GLY", "A", 2, "   0.000  -0.000   0.001", tail="  1.00  0.00           N  "),
        L(7, " CA ", "GLY", "A", 2, "9999.9999999.999-999.999"),
    ]) + "\n"
    files["syn_crlf.pdb"] = files["syn_numbers.pdb"].replace("\n", "\r\n")
    files["syn_no_newline_at_end.pdb"] = files["syn_numbers.pdb"].rstrip("\n")
    files["syn_short_line.pdb"] = L(1, " N  ", "ALA", "A", 1, "  11.104   6.134  -6.504") + "\n" + "ATOM      2  CA  ALA A   1      11.1\n"
    files["syn_bad_number.pdb"] = L(1, " N  ", "ALA", "A", 1, "  11.104   abc    -6.504") + "\n"
    files["syn_long_line.pdb"] = (L(1, " N  ", "ALA", "A", 1, "  11.104   6.134  -6.504", tail="  1.00  0.00           N  ") + "x" * 60 +
                                  "ATOM      9  CA  ALA A   1       1.000   2.000   3.000  1.00  0.00           C  \n" +
                                  L(2, " C  ", "ALA", "A", 1, "   4.000   5.000   6.000") + "\n")
    trunc = lambda line, n: line[:n]
    files["syn_truncated_columns.pdb"] = "\n".join([
        trunc(L(1, " N  ", "ALA", "A", 1, "  11.104   6.134  -6.504"), 54),   # nothing after the coordinates
        trunc(L(2, " H  ", "ALA", "A", 1, "  12.104   6.134  -6.504"), 60),   # hydrogen by name only, no element column
        trunc(L(3, "1HB ", "ALA", "A", 1, "  13.104   6.134  -6.504"), 77),   # one short of the element column
        L(4, " HA ", "ALA", "A", 1, "  14.104   6.134  -6.504", tail="  1.00  0.00              "),  # blank element: name decides
        L(5, "HG21", "ILE", "A", 2, "  15.104   6.134  -6.504", tail="  1.00  0.00              "),
        L(6, " CA ", "ILE", "A", 2, "  16.104   6.134  -6.504", tail="  1.00  0.00           C"),   # 77 + newline = 78
        L(7, "CD  ", "XYZ", "A", 3, "  17.104   6.134  -6.504", tail="  1.00  0.00              "),  # cadmium-like name
        L(8, " D1 ", "ALA", "A", 4, "  18.104   6.134  -6.504", tail="  1.00  0.00           D  "),
    ]) + "\n"
    files["syn_unknowns.pdb"] = "\n".join([
        L(1, " CA ", "ALA", "A", 1, "   1.000   2.000   3.000"),
        L(2, " XX ", "ALA", "A", 1, "   2.000   2.000   3.000", tail="  1.00  0.00           C  "),  # unknown atom, known element
        L(3, " CA ", "UNK", "A", 2, "   3.000   2.000   3.000", tail="  1.00  0.00           C  "),  # unknown residue
        L(4, "FE  ", "HEM", "A", 3, "   4.000   2.000   3.000", tail="  1.00  0.00          FE  ", rec="HETATM"),
        L(5, " Q1 ", "UNK", "A", 4, "   5.000   2.000   3.000", tail="  1.00  0.00           Q  "),  # no such element: radius 0
        L(6, " O  ", "HOH", "A", 5, "   6.000   2.000   3.000", tail="  1.00  0.00           O  ", rec="HETATM"),
        L(7, "SE  ", "MSE", "A", 6, "   7.000   2.000   3.000", tail="  1.00  0.00          SE  "),
        L(8, " Zn ", "ZN ", "A", 7, "   8.000   2.000   3.000", tail="  1.00  0.00          Zn  ", rec="HETATM"),  # lower case symbol
        L(9, " P  ", " DA", "B", 1, "   9.000   2.000   3.000", tail="  1.00  0.00           P  "),
        L(10, " C1'", "  A", "B", 2, "  10.000   2.000   3.000"),
    ]) + "\n"
    files["syn_altloc_icode_chain.pdb"] = "\n".join([
        L(1, " N  ", "SER", "A", 1, "   1.000   2.000   3.000", alt=" ", tail="  1.00  0.00           N  "),
        L(2, " CA ", "SER", "A", 1, "   2.000   2.000   3.000", alt="B"),    # first label seen wins
        L(3, " CA ", "SER", "A", 1, "   2.100   2.000   3.000", alt="A"),
        L(4, " CB ", "SER", "A", 1, "   3.000   2.000   3.000", alt="A"),
        L(5, " CB ", "SER", "A", 1, "   3.100   2.000   3.000", alt="B"),
        L(6, " OG ", "SER", "A", 1, "   4.000   2.000   3.000", alt=" ", tail="  1.00  0.00           O  "),
        L(7, " N  ", "GLY", "A", 1, "   5.000   2.000   3.000", icode="A", tail="  1.00  0.00           N  "),  # insertion code
        L(8, " N  ", "GLY", "B", 1, "   6.000   2.000   3.000", icode="A", tail="  1.00  0.00           N  "),  # same number, new chain
        L(9, " CA ", "GLY", "A", 1, "   7.000   2.000   3.000", icode="A"),                                          # back to chain A
        "TER",
        "MODEL        2",
        L(10, " N  ", "ALA", "A", 1, "   8.000   2.000   3.000", tail="  1.00  0.00           N  "),
        "ENDMDL",
        L(11, " N  ", "ALA", "A", 2, "   9.000   2.000   3.000", tail="  1.00  0.00           N  "),
    ]) + "\n"
    files["syn_only_hetatm.pdb"] = L(1, " O  ", "HOH", "A", 1, "   1.000   2.000   3.000", tail="  1.00  0.00           O  ", rec="HETATM") + "\n"
    return files


def main():
    out = {}
    for name, text in synthetic_files().items():
        dst = os.path.join(HERE, "pdb", name)
        with open(dst, "w", newline="") as fh:
            fh.write(text)
        out[name] = {str(o): reference_view(dst, o) for o in OPTION_SETS}
    for name in FILES:
        src, dst = os.path.join(DATA, name), os.path.join(HERE, "pdb", name)
        if name in CUT_AFTER_MODELS:
            seen, keep = 0, []
            for line in open(src):
                keep.append(line)
                if line.startswith("ENDMDL"):
                    seen += 1
                    if seen == CUT_AFTER_MODELS[name]:
                        break
            open(dst, "w").write("".join(keep))
        else:
            shutil.copyfile(src, dst)
        os.chmod(dst, 0o644)
        out[name] = {str(o): reference_view(dst, o) for o in OPTION_SETS}
    os.makedirs(os.path.join(HERE, "cif"), exist_ok=True)
    for name in CIF_FILES:
        dst = os.path.join(HERE, "cif", name)
        shutil.copyfile(os.path.join(DATA, name), dst)
        os.chmod(dst, 0o644)
        out[name] = {str(o): reference_view(dst, o) for o in CIF_OPTION_SETS}
    for name, text in synthetic_cifs().items():
        dst = os.path.join(HERE, "cif", name)
        with open(dst, "w", newline="") as fh:
            fh.write(text)
        out[name] = {str(o): reference_view(dst, o) for o in CIF_OPTION_SETS}
    with open(os.path.join(HERE, "ingest.json"), "w") as fh:
        json.dump(out, fh, indent=1, sort_keys=True)
    print({k: v["0"].get("n_atoms", "fail") for k, v in out.items()})


if __name__ == "__main__":
    main()
