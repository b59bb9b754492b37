"""Minimal VCF / ancestral-allele ingest for the U/Q path.

Restates what the reference asks scikit-allel for (sai/utils/utils.py:78-186, 389-410,
435-555) without depending on it: one pass over a plain or gzip VCF, the GT of the selected
samples only, the first ALT allele (``alt_number=1``), a region filter ``chrom[:start-end]``
(1-based, inclusive), missing alleles as -1, calls padded / truncated to the population's
ploidy (``numbers={"GT": ploidy}``), and the unphased dosage = sum over the ploidy axis
(utils.py:410) as int8 -- the layout the GPU path consumes.

Polarisation (utils.py:492-555): only sites listed in the ancestral-allele BED are kept, sites
whose ancestral allele is neither REF nor ALT are dropped, and where ALT is ancestral every
allele call a becomes |a - 1| (so a missing allele, -1, becomes 2 -- the reference's behaviour,
reproduced here on purpose).

This reader is host-side Python sized for test-scale inputs; a native tokenizer is the "next"
row of the scope table.
"""

from __future__ import annotations

import gzip
from dataclasses import dataclass
from typing import Optional, Sequence

import numpy as np


def _open_text(path: str):
    with open(path, "rb") as f:
        magic = f.read(2)
    if magic == b"\x1f\x8b":
        return gzip.open(path, "rt")
    return open(path, "r")


@dataclass
class VcfRegion:
    """Raw records of one chromosome region for a chosen set of samples."""

    samples: list[str]  # the selected samples, in the order requested
    pos: np.ndarray  # int32 [n_sites]
    ref: list[str]
    alt: list[str]  # first ALT allele
    gt: list[list[str]]  # [n_sites][n_samples] GT strings ("0|1", "./.", "1|0|1|0", ...)

    def __len__(self) -> int:
        return len(self.pos)


def first_last_pos(vcf_file: str, chr_name: str) -> tuple[Optional[int], Optional[int]]:
    """First and last POS of the first contiguous run of ``chr_name`` records
    (chunk_generator.py:64-73 scans the file the same way)."""
    first = last = None
    with _open_text(vcf_file) as f:
        for line in f:
            if line.startswith("#"):
                continue
            tab = line.find("\t")
            chrom = line[:tab]
            if chrom != chr_name:
                if first is not None:
                    break
                continue
            tab2 = line.find("\t", tab + 1)
            p = int(line[tab + 1 : tab2])
            if first is None:
                first = p
            last = p
    return first, last


def read_region(
    vcf_file: str, chr_name: str, samples: Sequence[str], start: Optional[int] = None, end: Optional[int] = None
) -> VcfRegion:
    """All records of ``chr_name`` with start <= POS <= end for the given samples."""
    chr_name = str(chr_name)
    header = None
    cols: list[int] = []
    pos, ref, alt, gts = [], [], [], []
    with _open_text(vcf_file) as f:
        for line in f:
            if line.startswith("##"):
                continue
            if line.startswith("#CHROM"):
                header = line.rstrip("\n").split("\t")
                names = header[9:]
                missing = [s for s in samples if s not in names]
                if missing:
                    raise ValueError(f"samples not found in {vcf_file}: {', '.join(missing)}")
                index = {n: i for i, n in enumerate(names)}
                cols = [9 + index[s] for s in samples]
                continue
            if header is None:
                raise ValueError(f"{vcf_file}: no #CHROM header line before the records")
            fields = line.rstrip("\n").split("\t")
            if fields[0] != chr_name:
                continue
            p = int(fields[1])
            if (start is not None and p < start) or (end is not None and p > end):
                continue
            fmt = fields[8].split(":")
            try:
                gi = fmt.index("GT")
            except ValueError:
                raise ValueError(f"{vcf_file}: record {fields[0]}:{p} has no GT field")
            pos.append(p)
            ref.append(fields[3])
            alt.append(fields[4].split(",")[0])
            if gi == 0:
                gts.append([fields[c].split(":", 1)[0] for c in cols])
            else:
                gts.append([fields[c].split(":")[gi] for c in cols])
    if header is None:
        raise ValueError(f"{vcf_file}: not a VCF (no #CHROM header)")
    return VcfRegion(list(samples), np.array(pos, dtype=np.int32), ref, alt, gts)


def read_site_columns(vcf_file: str, chr_name: str, start: Optional[int] = None, end: Optional[int] = None):
    """(POS int32, REF, first ALT) of the records of ``chr_name`` with start <= POS <= end: the fixed columns only,
    no sample column is looked at (the allele-level reader takes the calls from the native tokenizer)."""
    chr_name = str(chr_name)
    pos, ref, alt = [], [], []
    with _open_text(vcf_file) as f:
        for line in f:
            if line.startswith("#"):
                continue
            fields = line.split("\t", 5)
            if fields[0] != chr_name:
                continue
            p = int(fields[1])
            if (start is not None and p < start) or (end is not None and p > end):
                continue
            pos.append(p)
            ref.append(fields[3])
            alt.append(fields[4].split(",")[0])
    return np.array(pos, dtype=np.int32), ref, alt


def _alleles(gt: str, ploidy: int) -> list[int]:
    """Allele indices of one call, -1 for '.', padded with -1 / cut to ``ploidy``."""
    parts = gt.replace("|", "/").split("/")
    out = []
    for a in parts[:ploidy]:
        out.append(-1 if a == "." or a == "" else int(a))
    out.extend([-1] * (ploidy - len(out)))
    return out


def dosage_matrices(region: VcfRegion, columns: Sequence[int], ploidy: int) -> tuple[np.ndarray, np.ndarray]:
    """(dosage, flipped_dosage), both int8 [n_sites][len(columns)].

    dosage = sum of the allele calls (utils.py:410); flipped_dosage = sum of |a - 1|, what the
    same call sums to after ``flip_snps`` (utils.py:539-555)."""
    n = len(region)
    dos = np.zeros((n, len(columns)), dtype=np.int16)
    fdos = np.zeros((n, len(columns)), dtype=np.int16)
    cache: dict[str, tuple[int, int]] = {}
    for i, row in enumerate(region.gt):
        for j, c in enumerate(columns):
            g = row[c]
            v = cache.get(g)
            if v is None:
                al = _alleles(g, ploidy)
                v = (sum(al), sum(abs(a - 1) for a in al))
                cache[g] = v
            dos[i, j], fdos[i, j] = v
    if n and (dos.max() > 127 or fdos.max() > 127 or dos.min() < -128):
        raise ValueError("dosage outside the int8 range")
    return dos.astype(np.int8), fdos.astype(np.int8)


def read_anc_allele(anc_allele_file: str, chr_name: str, start: int = None, end: int = None) -> dict[str, dict[int, str]]:
    """BED with columns chrom, start, pos, allele -> {chrom: {pos: allele}} restricted to the
    chromosome and (inclusive) region; no entry left is a ValueError (utils.py:435-489)."""
    out: dict[str, dict[int, str]] = {}
    try:
        with open(anc_allele_file, "r") as f:
            for line in f:
                e = line.rstrip().split()
                if not e:
                    continue
                chrom, p, allele = e[0], int(e[2]), e[3]
                if chrom != chr_name:
                    continue
                if (start is not None and p < start) or (end is not None and p > end):
                    continue
                out.setdefault(chrom, {})[p] = allele
    except FileNotFoundError as exc:
        raise FileNotFoundError(f"File {anc_allele_file} not found.") from exc
    if not out:
        if start is not None or end is not None:
            raise ValueError(
                f"No ancestral allele is found for chromosome {chr_name} in the region {start}-{end}."
            )
        raise ValueError(f"No ancestral allele is found for chromosome {chr_name}.")
    return out


def polarisation_masks(region: VcfRegion, anc: dict[int, str]) -> tuple[np.ndarray, np.ndarray]:
    """(keep, flip) boolean masks over the region's sites (utils.py:511-537): keep = listed in
    the BED and ancestral allele is REF or ALT; flip = ancestral allele is ALT."""
    n = len(region)
    keep = np.zeros(n, dtype=bool)
    flip = np.zeros(n, dtype=bool)
    for i in range(n):
        a = anc.get(int(region.pos[i]))
        if a is None:
            continue
        if a == region.alt[i]:
            keep[i] = flip[i] = True
        elif a == region.ref[i]:
            keep[i] = True
    return keep, flip
