"""Python face of include/freesasa_ingest.h: multi-threaded PDB / mmCIF -> packed batch.

    batch = ingest.load_pdb_files(paths)            # host threads, one structure per file
    sasa, totals = freesasa_amd.calc_batch(batch.xyz, batch.radii, batch.offsets)
    per_res = batch.residue_sums(sasa)              # or GpuContext.segment_sums on the device

What a file contributes is what the reference's freesasa_structure_from_pdb() holds for it
(ref: src/structure.c:644-722); the parsing itself is C (freesasa_amd/csrc/ingest.c)."""
import ctypes as C

import numpy as np

from . import lib

INCLUDE_HETATM, INCLUDE_HYDROGEN, JOIN_MODELS = 1, 1 << 2, 1 << 5     # ref: src/freesasa.h:182-191
HALT_AT_UNKNOWN, SKIP_UNKNOWN, RADIUS_FROM_OCCUPANCY = 1 << 6, 1 << 7, 1 << 8
PARSE_ON_DEVICE = 1 << 16       # sweep drivers only: the files' text is parsed by kernels (csrc/gpu_parse.hip)
OK, EIO, EFORMAT, EEMPTY, EUNKNOWN, EOPTION, ENOMEM, EVERSION = range(8)
APOLAR, POLAR, UNKNOWN = 0, 1, 2


class _CBatch(C.Structure):
    _fields_ = [("n_structs", C.c_int32), ("n_atoms", C.c_int64), ("n_residues", C.c_int64),
                ("xyz", C.POINTER(C.c_double)), ("radii", C.POINTER(C.c_double)),
                ("atom_class", C.POINTER(C.c_uint8)), ("atom_backbone", C.POINTER(C.c_uint8)),
                ("atom_name", C.POINTER(C.c_char)), ("atom_symbol", C.POINTER(C.c_char)),
                ("offsets", C.POINTER(C.c_int64)),
                ("res_first", C.POINTER(C.c_int64)), ("res_offsets", C.POINTER(C.c_int64)),
                ("res_ref", C.POINTER(C.c_int16)), ("res_name", C.POINTER(C.c_char)), ("res_number", C.POINTER(C.c_char)),
                ("res_chain", C.POINTER(C.c_char)), ("status", C.POINTER(C.c_int32))]


def _arr(ptr, n, dtype):
    """numpy copy of n items behind a ctypes pointer (one memcpy; np.ctypeslib.as_array is slow on large arrays)"""
    if n == 0:
        return np.zeros(0, dtype=dtype)
    nbytes = n * np.dtype(dtype).itemsize
    view = (C.c_char * nbytes).from_address(C.addressof(ptr.contents))
    return np.frombuffer(view, dtype=dtype).copy()


class Batch:
    """numpy copy of a freesasa_ingest_batch (field meanings: include/freesasa_ingest.h)."""

    def __init__(self, cb):
        na, nr, ns = cb.n_atoms, cb.n_residues, cb.n_structs
        self.n_structs, self.n_atoms, self.n_residues = ns, na, nr
        self.xyz = _arr(cb.xyz, 3 * na, np.float64).reshape(-1, 3)
        self.radii = _arr(cb.radii, na, np.float64)
        self.atom_class = _arr(cb.atom_class, na, np.uint8)
        self.atom_backbone = _arr(cb.atom_backbone, na, np.uint8)
        self.res_ref = _arr(cb.res_ref, nr, np.int16)
        self.atom_name_raw = np.frombuffer(C.string_at(cb.atom_name, 4 * na) if na else b"", dtype="S4").copy()
        self.atom_symbol_raw = np.frombuffer(C.string_at(cb.atom_symbol, 2 * na) if na else b"", dtype="S2").copy()
        self.offsets = _arr(cb.offsets, ns + 1, np.int64)
        self.res_first = _arr(cb.res_first, nr + 1, np.int64)
        self.res_offsets = _arr(cb.res_offsets, ns + 1, np.int64)
        self.status = _arr(cb.status, ns, np.int32)
        # residue labels stay fixed-width byte arrays until somebody asks for strings
        self.res_name_raw = np.frombuffer(C.string_at(cb.res_name, 4 * nr) if nr else b"", dtype="S4").copy()
        self.res_number_raw = np.frombuffer(C.string_at(cb.res_number, 6 * nr) if nr else b"", dtype="S6").copy()
        self.res_chain_raw = np.frombuffer(C.string_at(cb.res_chain, 4 * nr) if nr else b"", dtype="S4").copy()

    @property
    def res_name(self):
        return [v.decode() for v in self.res_name_raw.tolist()]

    @property
    def res_number(self):
        return [v.decode() for v in self.res_number_raw.tolist()]

    @property
    def res_chain(self):
        return [v.decode() for v in self.res_chain_raw.tolist()]

    def _as_c(self):
        """A freesasa_ingest_batch view of the numpy arrays (no copies; valid while self lives)."""
        cb = _CBatch()
        cb.n_structs, cb.n_atoms, cb.n_residues = self.n_structs, self.n_atoms, self.n_residues
        ptr = lambda a, t: C.cast(a.ctypes.data, C.POINTER(t))
        cb.offsets, cb.res_first, cb.res_offsets = ptr(self.offsets, C.c_int64), ptr(self.res_first, C.c_int64), ptr(self.res_offsets, C.c_int64)
        cb.atom_name, cb.atom_symbol = ptr(self.atom_name_raw, C.c_char), ptr(self.atom_symbol_raw, C.c_char)
        cb.res_name, cb.res_number, cb.res_chain = ptr(self.res_name_raw, C.c_char), ptr(self.res_number_raw, C.c_char), ptr(self.res_chain_raw, C.c_char)
        self._xyz_flat = np.ascontiguousarray(self.xyz, dtype=np.float64).reshape(-1)
        cb.xyz, cb.radii = ptr(self._xyz_flat, C.c_double), ptr(self.radii, C.c_double)
        cb.atom_class, cb.atom_backbone = ptr(self.atom_class, C.c_uint8), ptr(self.atom_backbone, C.c_uint8)
        cb.res_ref, cb.status = ptr(self.res_ref, C.c_int16), ptr(self.status, C.c_int32)
        return cb

    def save(self, path):
        """freesasa_ingest_save(): the batch as a binary cache file (load_cache() reads it back)."""
        cb = self._as_c()
        rc = _proto().freesasa_ingest_save(C.byref(cb), str(path).encode())
        if rc:
            raise RuntimeError(f"freesasa_ingest_save failed with code {rc}")

    def select(self, structure, command):
        """The reference's selection language on one structure: (name, mask[n_atoms of it], warned).
        Raises ValueError on a syntax error (the reference returns FREESASA_FAIL)."""
        L = _proto()
        n = int(self.offsets[structure + 1] - self.offsets[structure])
        mask = np.zeros(n, dtype=np.uint8)
        name = C.create_string_buffer(64)
        cb = self._as_c()
        ret = L.freesasa_ingest_select(C.byref(cb), structure, command.encode(), name, mask.ctypes.data_as(C.POINTER(C.c_ubyte)))
        if ret == -1:
            raise ValueError(f"cannot parse selection {command!r}")
        return name.value.decode(), mask, ret == -2

    def residue_sums(self, per_atom):
        """Host-side segmented sum over the residues (the device-side one is GpuContext.segment_sums)."""
        per_atom = np.asarray(per_atom, dtype=np.float64)
        if self.n_residues == 0:
            return np.zeros(0)
        return np.add.reduceat(np.append(per_atom, 0.0), self.res_first[:-1])


def _proto():
    L = lib()
    if not getattr(L, "_ingest_ready", False):
        L.freesasa_ingest_pdb_files.argtypes = [C.POINTER(C.c_char_p), C.c_int, C.c_int, C.c_int, C.POINTER(_CBatch)]
        L.freesasa_ingest_pdb_texts.argtypes = [C.POINTER(C.c_char_p), C.POINTER(C.c_size_t), C.c_int, C.c_int, C.c_int,
                                                C.POINTER(_CBatch)]
        L.freesasa_ingest_free.argtypes = [C.POINTER(_CBatch)]
        L.freesasa_ingest_free.restype = None
        L.freesasa_ingest_protor_radius.argtypes = [C.c_char_p, C.c_char_p, C.POINTER(C.c_int)]
        L.freesasa_ingest_protor_radius.restype = C.c_double
        L.freesasa_ingest_guess_radius.argtypes = [C.c_char_p]
        L.freesasa_ingest_guess_radius.restype = C.c_double
        L.freesasa_ingest_residue_reference_table.argtypes = [C.POINTER(C.c_double)]
        L.freesasa_ingest_is_backbone.argtypes = [C.c_char_p]
        L.freesasa_ingest_select.argtypes = [C.POINTER(_CBatch), C.c_int, C.c_char_p, C.c_char_p, C.POINTER(C.c_ubyte)]
        L.freesasa_ingest_save.argtypes = [C.POINTER(_CBatch), C.c_char_p]
        L.freesasa_ingest_load.argtypes = [C.c_char_p, C.POINTER(_CBatch)]
        L.freesasa_ingest_load_mt.argtypes = [C.c_char_p, C.c_int, C.POINTER(_CBatch)]
        L.freesasa_ingest_usable_cpus.restype = C.c_int
        L.freesasa_ingest_trim.argtypes = [C.c_size_t]
        L.freesasa_ingest_trim.restype = C.c_size_t
        L.freesasa_ingest_cache_open.argtypes = [C.c_char_p, C.POINTER(C.c_void_p)]
        L.freesasa_ingest_cache_close.argtypes = [C.c_void_p]
        L.freesasa_ingest_cache_n_structs.argtypes = [C.c_void_p]; L.freesasa_ingest_cache_n_structs.restype = C.c_int32
        L.freesasa_ingest_cache_n_atoms.argtypes = [C.c_void_p]; L.freesasa_ingest_cache_n_atoms.restype = C.c_int64
        L.freesasa_ingest_cache_offsets.argtypes = [C.c_void_p]; L.freesasa_ingest_cache_offsets.restype = C.POINTER(C.c_int64)
        L.freesasa_ingest_cache_status.argtypes = [C.c_void_p]; L.freesasa_ingest_cache_status.restype = C.POINTER(C.c_int32)
        L.freesasa_ingest_cache_read_atoms.argtypes = [C.c_void_p, C.c_int64, C.c_int64, C.POINTER(C.c_double), C.POINTER(C.c_double), C.POINTER(C.c_ubyte)]
        L._ingest_ready = True
    return L


def _finish(L, rc, cb):
    if rc:
        raise RuntimeError(f"freesasa_ingest failed with code {rc}")
    try:
        return Batch(cb)
    finally:
        L.freesasa_ingest_free(C.byref(cb))


def load_pdb_files(paths, options=0, n_threads=0):
    """Read PDB files into one Batch; per-file failures are in batch.status (empty structures)."""
    L = _proto()
    arr = (C.c_char_p * len(paths))(*[str(p).encode() for p in paths])
    cb = _CBatch()
    return _finish(L, L.freesasa_ingest_pdb_files(arr, len(paths), options, n_threads, C.byref(cb)), cb)


def load_cache(path, n_threads=0):
    """freesasa_ingest_load_mt(): a Batch from the binary cache file Batch.save() wrote, its 1 MiB pieces read and
    verified by n_threads readers (<= 0: the usable CPUs, at most 8).  Raises RuntimeError with the library's code
    (EIO: cannot open; EFORMAT: not a cache file, truncated, checksum or offsets wrong)."""
    L = _proto()
    cb = _CBatch()
    return _finish(L, L.freesasa_ingest_load_mt(str(path).encode(), int(n_threads), C.byref(cb)), cb)


def trim(keep_bytes=0):
    """freesasa_ingest_trim(): give the loader's kept blocks (freed batches, kept for the next ones: at most 1 GiB) back
    to the allocator until at most keep_bytes remain; returns the bytes released."""
    return int(_proto().freesasa_ingest_trim(int(keep_bytes)))


def usable_cpus():
    """freesasa_ingest_usable_cpus(): the affinity mask capped by the cgroup's CPU quota."""
    return int(_proto().freesasa_ingest_usable_cpus())


class Cache:
    """A cache file read partially (freesasa_ingest_cache_*): offsets and status of every structure, and
    read_atoms(a0, a1) -> (xyz, radii, atom_class) of a run of atoms, verified piece by piece."""

    def __init__(self, path):
        L = _proto()
        self._h = C.c_void_p()
        rc = L.freesasa_ingest_cache_open(str(path).encode(), C.byref(self._h))
        if rc:
            raise RuntimeError(f"freesasa_ingest_cache_open failed with code {rc}")
        self.n_structs = int(L.freesasa_ingest_cache_n_structs(self._h))
        self.n_atoms = int(L.freesasa_ingest_cache_n_atoms(self._h))
        self.offsets = np.ctypeslib.as_array(L.freesasa_ingest_cache_offsets(self._h), (self.n_structs + 1,)).copy()
        self.status = (np.ctypeslib.as_array(L.freesasa_ingest_cache_status(self._h), (self.n_structs,)).copy()
                       if self.n_structs else np.zeros(0, np.int32))

    def read_atoms(self, a0, a1):
        n = int(a1) - int(a0)
        xyz, r, cls = np.empty((max(n, 0), 3)), np.empty(max(n, 0)), np.empty(max(n, 0), np.uint8)
        rc = _proto().freesasa_ingest_cache_read_atoms(self._h, int(a0), int(a1), xyz.ctypes.data_as(C.POINTER(C.c_double)),
                                                       r.ctypes.data_as(C.POINTER(C.c_double)), cls.ctypes.data_as(C.POINTER(C.c_ubyte)))
        if rc:
            raise RuntimeError(f"freesasa_ingest_cache_read_atoms failed with code {rc}")
        return xyz, r, cls

    def close(self):
        if self._h:
            _proto().freesasa_ingest_cache_close(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def load_pdb_texts(texts, options=0, n_threads=0):
    """Same for PDB texts in memory (bytes or str)."""
    L = _proto()
    raw = [t.encode() if isinstance(t, str) else bytes(t) for t in texts]
    arr = (C.c_char_p * len(raw))(*raw)
    lens = (C.c_size_t * len(raw))(*[len(t) for t in raw])
    cb = _CBatch()
    return _finish(L, L.freesasa_ingest_pdb_texts(arr, lens, len(raw), options, n_threads, C.byref(cb)), cb)


def residue_reference_table():
    """[rows, 5] reference areas (total, main chain, side chain, polar, apolar) that Batch.res_ref indexes."""
    L = _proto()
    n = L.freesasa_ingest_residue_reference_table(None)
    t = np.empty(5 * n)
    L.freesasa_ingest_residue_reference_table(t.ctypes.data_as(C.POINTER(C.c_double)))
    return t.reshape(n, 5)


load_files, load_texts = load_pdb_files, load_pdb_texts      # the format is recognised per input (PDB or mmCIF)


def protor_radius(res_name, atom_name):
    """(radius, class) of the ProtOr classifier; radius -1.0 and class UNKNOWN if the atom is not known."""
    cls = C.c_int()
    r = _proto().freesasa_ingest_protor_radius(res_name.encode(), atom_name.encode(), C.byref(cls))
    return r, cls.value


def guess_radius(symbol):
    return _proto().freesasa_ingest_guess_radius(symbol.encode())
