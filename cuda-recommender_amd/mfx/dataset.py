"""Rating-matrix containers, the reference's binary dataset-directory format, and a seeded
synthetic generator.

The reference ships no dataset and no converter; its loader defines the format only
implicitly (reference: src/tools.cpp:3-85 `load`, src/pmf_util.h:108-136 and :171-194
`read_binary_file`, src/extras.cpp:24-44 `generate_file_pointers`).  This module is the
host-side writer/reader for that format plus the generator SURVEY.md §8(d) asks for.

Layout of a dataset directory (all binary files little-endian, indices 0-based):

    meta_modified_all   text: "m n nnz" / 3 COO names (parsed, never opened) /
                        "row_ptr col_idx csr_val" names / "col_ptr row_idx csc_val" names /
                        "nnz_test" / "test_val test_row test_col" names
    *_ptr files         int32  [rows+1] / [cols+1]
    *_idx files         uint32 [nnz]
    *_val files         float32 [nnz]
    meta                text: "m n" / "nnz train_name" / "nnz_test test_name"
"""
from __future__ import annotations

import os
from dataclasses import dataclass, field
from typing import Optional

import numpy as np

_NAMES = dict(
    coo_val="R_train_coo.data.bin", coo_row="R_train_coo.row.bin", coo_col="R_train_coo.col.bin",
    csr_ptr="R_train_csr.indptr.bin", csr_idx="R_train_csr.indices.bin", csr_val="R_train_csr.data.bin",
    csc_ptr="R_train_csc.indptr.bin", csc_idx="R_train_csc.indices.bin", csc_val="R_train_csc.data.bin",
    test_val="R_test_coo.data.bin", test_row="R_test_coo.row.bin", test_col="R_test_coo.col.bin",
    train_txt="train.ratings", test_txt="test.ratings",
)


@dataclass
class RatingData:
    """Dual CSR+CSC training matrix + COO test set (reference: SparseMatrix + TestData,
    src/pmf_util.h:34-211).  Entry order inside a row/column is the summation order."""
    rows: int
    cols: int
    csr_row_ptr: np.ndarray
    csr_col_idx: np.ndarray
    csr_val: np.ndarray
    csc_col_ptr: np.ndarray
    csc_row_idx: np.ndarray
    csc_val: np.ndarray
    test_row: np.ndarray = field(default_factory=lambda: np.zeros(0, np.uint32))
    test_col: np.ndarray = field(default_factory=lambda: np.zeros(0, np.uint32))
    test_val: np.ndarray = field(default_factory=lambda: np.zeros(0, np.float32))

    @property
    def nnz(self) -> int:
        return int(self.csr_val.shape[0])

    @property
    def nnz_test(self) -> int:
        return int(self.test_val.shape[0])

    def copy(self) -> "RatingData":
        return RatingData(self.rows, self.cols, *[a.copy() for a in (
            self.csr_row_ptr, self.csr_col_idx, self.csr_val, self.csc_col_ptr, self.csc_row_idx,
            self.csc_val, self.test_row, self.test_col, self.test_val)])

    def check_types(self) -> None:
        """What the ctypes boundary needs to be memory-safe (shapes, dtypes); the CONTENT checks -- monotone
        pointers, indices in range -- are the library's (MFX_ERR_INVALID), not repeated here."""
        z = self.nnz
        if self.csr_row_ptr.shape != (self.rows + 1,) or self.csc_col_ptr.shape != (self.cols + 1,):
            raise ValueError("pointer arrays must have rows + 1 / cols + 1 entries")
        for a in (self.csr_col_idx, self.csc_row_idx, self.csr_val, self.csc_val):
            if a.shape != (z,):
                raise ValueError("index / value arrays must have nnz entries")
        if not (self.test_row.shape == self.test_col.shape == self.test_val.shape):
            raise ValueError("test COO arrays must have equal lengths")
        for a in (self.csr_row_ptr, self.csr_col_idx, self.csc_col_ptr, self.csc_row_idx, self.test_row, self.test_col):
            if a.dtype != np.uint32 or not a.flags.c_contiguous:
                raise ValueError("index arrays must be contiguous uint32")
        for a in (self.csr_val, self.csc_val, self.test_val):
            if a.dtype != np.float32 or not a.flags.c_contiguous:
                raise ValueError("value arrays must be contiguous float32")

    def validate(self) -> None:
        z = self.nnz
        assert self.csr_row_ptr.shape == (self.rows + 1,) and self.csc_col_ptr.shape == (self.cols + 1,)
        assert self.csr_row_ptr[0] == 0 and self.csr_row_ptr[-1] == z
        assert self.csc_col_ptr[0] == 0 and self.csc_col_ptr[-1] == z
        assert self.csr_col_idx.shape == (z,) and self.csc_row_idx.shape == (z,) and self.csc_val.shape == (z,)
        if z:
            assert int(self.csr_col_idx.max()) < self.cols and int(self.csc_row_idx.max()) < self.rows
        assert np.all(np.diff(self.csr_row_ptr.astype(np.int64)) >= 0)
        assert np.all(np.diff(self.csc_col_ptr.astype(np.int64)) >= 0)
        for a in (self.csr_row_ptr, self.csr_col_idx, self.csc_col_ptr, self.csc_row_idx, self.test_row, self.test_col):
            assert a.dtype == np.uint32
        for a in (self.csr_val, self.csc_val, self.test_val):
            assert a.dtype == np.float32


def from_coo(rows: int, cols: int, r: np.ndarray, c: np.ndarray, v: np.ndarray,
             test_r: Optional[np.ndarray] = None, test_c: Optional[np.ndarray] = None,
             test_v: Optional[np.ndarray] = None) -> RatingData:
    """Builds both orientations from COO triplets (duplicates are the caller's problem).
    CSR is ordered by (row, col), CSC by (col, row) -- a stable, documented summation order."""
    r = np.asarray(r, np.int64); c = np.asarray(c, np.int64); v = np.asarray(v, np.float32)
    o = np.lexsort((c, r))
    csr_ptr = np.zeros(rows + 1, np.int64); np.add.at(csr_ptr, r + 1, 1); csr_ptr = np.cumsum(csr_ptr)
    o2 = np.lexsort((r, c))
    csc_ptr = np.zeros(cols + 1, np.int64); np.add.at(csc_ptr, c + 1, 1); csc_ptr = np.cumsum(csc_ptr)
    u32 = lambda a: np.ascontiguousarray(a, dtype=np.uint32)
    z = lambda: np.zeros(0, np.uint32)
    return RatingData(
        rows, cols, u32(csr_ptr), u32(c[o]), np.ascontiguousarray(v[o]),
        u32(csc_ptr), u32(r[o2]), np.ascontiguousarray(v[o2]),
        u32(test_r) if test_r is not None else z(), u32(test_c) if test_c is not None else z(),
        np.ascontiguousarray(test_v, dtype=np.float32) if test_v is not None else np.zeros(0, np.float32))


def synth_ratings(rows: int, cols: int, nnz: int, seed: int = 1234, skew: float = 1.0,
                  planted_rank: int = 8, noise: float = 0.1, test_frac: float = 0.01,
                  empty_row_frac: float = 0.0, empty_col_frac: float = 0.0) -> RatingData:
    """Seeded synthetic ratings (SURVEY.md §8d "Synthetic inputs"): `nnz` distinct (i, j) with
    Zipf-like row/column popularity (`skew`=0 is uniform), ratings = planted rank-`planted_rank`
    model mapped to [1, 5] plus N(0, noise^2), clipped; `test_frac` of the draws held out as
    the test set.  `empty_*_frac` forces that share of rows/columns to have no rating at all
    (the reference's zero-row rule, src/CCD.cpp:8 / src/ALS.cpp:151-157)."""
    rng = np.random.default_rng(seed)

    def weights(n, empty_frac):
        w = 1.0 / np.power(np.arange(1, n + 1, dtype=np.float64), skew)
        rng.shuffle(w)
        if empty_frac > 0:
            w[rng.choice(n, size=max(1, int(n * empty_frac)), replace=False)] = 0.0
        return np.cumsum(w / w.sum())

    cw_r, cw_c = weights(rows, empty_row_frac), weights(cols, empty_col_frac)
    want = nnz + int(round(nnz * test_frac))
    if want > 0.6 * rows * cols:
        raise ValueError("requested density too high for rejection sampling")
    keys = np.zeros(0, np.int64)
    while keys.shape[0] < want:
        need = int((want - keys.shape[0]) * 1.3) + 16
        i = np.minimum(np.searchsorted(cw_r, rng.random(need)), rows - 1).astype(np.int64)
        j = np.minimum(np.searchsorted(cw_c, rng.random(need)), cols - 1).astype(np.int64)
        keys = np.unique(np.concatenate([keys, i * cols + j]))
    keys = rng.permutation(keys)[:want]
    i, j = keys // cols, keys % cols
    ws = rng.standard_normal((rows, planted_rank)).astype(np.float32)
    hs = rng.standard_normal((cols, planted_rank)).astype(np.float32)
    raw = np.einsum("ij,ij->i", ws[i], hs[j]) / np.sqrt(planted_rank)
    val = 3.0 + 1.0 * raw + noise * rng.standard_normal(want)
    val = np.clip(val, 1.0, 5.0).astype(np.float32)
    return from_coo(rows, cols, i[:nnz], j[:nnz], val[:nnz], i[nnz:], j[nnz:], val[nnz:])


def write_dataset_dir(path: str, d: RatingData) -> None:
    """Writes `d` in the reference's dataset-directory format (see module docstring)."""
    os.makedirs(path, exist_ok=True)
    p = lambda k: os.path.join(path, _NAMES[k])
    # COO triplets: named in meta_modified_all (src/tools.cpp:30-35) but never opened there.
    rr = np.repeat(np.arange(d.rows, dtype=np.uint32), np.diff(d.csr_row_ptr.astype(np.int64)))
    d.csr_val.astype("<f4").tofile(p("coo_val")); rr.astype("<u4").tofile(p("coo_row")); d.csr_col_idx.astype("<u4").tofile(p("coo_col"))
    d.csr_row_ptr.astype("<i4").tofile(p("csr_ptr")); d.csr_col_idx.astype("<u4").tofile(p("csr_idx")); d.csr_val.astype("<f4").tofile(p("csr_val"))
    d.csc_col_ptr.astype("<i4").tofile(p("csc_ptr")); d.csc_row_idx.astype("<u4").tofile(p("csc_idx")); d.csc_val.astype("<f4").tofile(p("csc_val"))
    d.test_val.astype("<f4").tofile(p("test_val")); d.test_row.astype("<u4").tofile(p("test_row")); d.test_col.astype("<u4").tofile(p("test_col"))
    with open(os.path.join(path, "meta_modified_all"), "w") as f:
        f.write(f"{d.rows} {d.cols} {d.nnz}\n")
        f.write(f"{_NAMES['coo_val']} {_NAMES['coo_row']} {_NAMES['coo_col']}\n")
        f.write(f"{_NAMES['csr_ptr']} {_NAMES['csr_idx']} {_NAMES['csr_val']}\n")
        f.write(f"{_NAMES['csc_ptr']} {_NAMES['csc_idx']} {_NAMES['csc_val']}\n")
        f.write(f"{d.nnz_test}\n")
        f.write(f"{_NAMES['test_val']} {_NAMES['test_row']} {_NAMES['test_col']}\n")
    # `meta` + 1-based text files: what generate_file_pointers/open_files expect to exist
    # (src/extras.cpp:3-44); text indices are 1-based like calculate_rmse_from_file's (:169).
    with open(os.path.join(path, "meta"), "w") as f:
        f.write(f"{d.rows} {d.cols}\n{d.nnz} {_NAMES['train_txt']}\n{d.nnz_test} {_NAMES['test_txt']}\n")
    with open(os.path.join(path, _NAMES["test_txt"]), "w") as f:
        for a, b, c in zip(d.test_row, d.test_col, d.test_val):
            f.write(f"{int(a) + 1} {int(b) + 1} {float(c):.6g}\n")


def read_dataset_dir(path: str) -> RatingData:
    """Reads a dataset directory exactly as the reference's `load` does (src/tools.cpp:3-85):
    only meta_modified_all and the nine CSR/CSC/test binaries are touched."""
    with open(os.path.join(path, "meta_modified_all")) as f:
        tok = f.read().split()
    if len(tok) < 16:
        raise ValueError("meta_modified_all: expected 16 whitespace-separated tokens")
    m, n, nnz = int(tok[0]), int(tok[1]), int(tok[2])
    csr_ptr, csr_idx, csr_val, csc_ptr, csc_idx, csc_val = tok[6:12]
    nnz_test = int(tok[12]); tv, tr, tc = tok[13:16]
    rd = lambda name, dt, cnt: np.fromfile(os.path.join(path, name), dtype=dt, count=cnt)
    d = RatingData(
        m, n,
        rd(csr_ptr, "<i4", m + 1).astype(np.uint32), rd(csr_idx, "<u4", nnz), rd(csr_val, "<f4", nnz),
        rd(csc_ptr, "<i4", n + 1).astype(np.uint32), rd(csc_idx, "<u4", nnz), rd(csc_val, "<f4", nnz),
        rd(tr, "<u4", nnz_test), rd(tc, "<u4", nnz_test), rd(tv, "<f4", nnz_test))
    d.validate()
    return d


def _read_triplets(pth: str, sep: Optional[str]):
    """(row, col, rating) columns of a ratings text file.  Formats met in the wild for the datasets the reference's
    scripts name (scripts/times.sh:15-35): `i j r` (LIBPMF), `user<TAB>item<TAB>rating<TAB>timestamp` (ML-100K u.data),
    `user::item::rating::timestamp` (ML-1M / ML-10M ratings.dat), `userId,movieId,rating,timestamp` under a header line
    (ML-20M ratings.csv).  Further columns are ignored; `sep` None = detect."""
    with open(pth) as f:
        first = f.readline()
        tokens = first.replace("::", " ").replace(",", " ").split()
        header = bool(tokens) and not tokens[0].lstrip("+-").replace(".", "", 1).isdigit()
        if sep is None:
            sep = "::" if "::" in first else ("," if "," in first else None)
        rest = f.read()
    text = rest if header else first + rest
    if sep == "::":  # numpy's reader wants a one-character delimiter
        text, sep = text.replace("::", " "), None
    import io
    a = np.loadtxt(io.StringIO(text), delimiter=sep, usecols=(0, 1, 2), dtype=np.float64, ndmin=2)
    return a[:, 0].astype(np.int64), a[:, 1].astype(np.int64), a[:, 2].astype(np.float32)


def convert_text_ratings(train_path: str, out_dir: str, test_path: Optional[str] = None,
                         one_based: bool = True, sep: Optional[str] = None, compact_ids: bool = False,
                         test_frac: float = 0.0, seed: int = 0) -> RatingData:
    """Converter for ratings text files (see _read_triplets for the formats) into the binary directory format -- the
    tool the reference's authors used but never shipped.  compact_ids: renumber the row / column ids that
    occur (train and test together) 0 .. n-1 in ascending order of id -- MovieLens item ids have gaps -- and write the
    original ids to row_ids.txt / col_ids.txt next to the dataset.  test_frac > 0 (and no test file): that fraction of
    the ratings, drawn with `seed`, becomes the test set -- the distributed files are not split."""
    off = 1 if one_based else 0
    r, c, v = _read_triplets(train_path, sep)
    tr = tc = tv = None
    if test_path:
        tr, tc, tv = _read_triplets(test_path, sep)
    elif test_frac > 0.0:
        held = np.random.default_rng(seed).random(r.size) < test_frac
        tr, tc, tv = r[held], c[held], v[held]
        r, c, v = r[~held], c[~held], v[~held]
    if compact_ids:
        rid = np.unique(np.concatenate([r, tr]) if tr is not None else r)
        cid = np.unique(np.concatenate([c, tc]) if tc is not None else c)
        r, c = np.searchsorted(rid, r), np.searchsorted(cid, c)
        if tr is not None:
            tr, tc = np.searchsorted(rid, tr), np.searchsorted(cid, tc)
        rows, cols = int(rid.size), int(cid.size)
    else:
        r, c = r - off, c - off
        if tr is not None:
            tr, tc = tr - off, tc - off
        if r.size and (r.min() < 0 or c.min() < 0):
            raise ValueError("negative index after the one_based shift: is the file 0-based?")
        rows = int(max(r.max(initial=-1), tr.max(initial=-1) if tr is not None else -1)) + 1
        cols = int(max(c.max(initial=-1), tc.max(initial=-1) if tc is not None else -1)) + 1
    d = from_coo(rows, cols, r, c, v, tr, tc, tv)
    write_dataset_dir(out_dir, d)
    if compact_ids:
        np.savetxt(os.path.join(out_dir, "row_ids.txt"), rid, fmt="%d")
        np.savetxt(os.path.join(out_dir, "col_ids.txt"), cid, fmt="%d")
    return d
