"""The allele-level half of ``sai.utils`` (sai/utils/utils.py:78-212, 359-432, 492-555): population
blocks with REF / ALT strings and per-allele calls ``GT[sites][individuals][ploidy]`` (int8, -1 = missing;
the array scikit-allel's ``GenotypeArray`` wraps), and the filters the reference applies to them.

Nothing on the ``score`` path comes through here: ``WindowGenerator`` asks ``read_data`` for unphased
dosages without filters (window_generator.py:105-120), which the native tokenizers serve
(``read_data.read_dosage_data`` / ``read_data_device``).  This module is for plug-in authors who use the
reference's reader options -- phased haplotype columns, fixed-variant and missing-call filters.  The calls
come from the native tokenizer all the same (``allele_calls``: the dosage of a call's first k alleles is what
``sai_vcf_load`` returns at ploidy k, so allele k is the difference of two reads); REF / ALT are the fixed
columns of the text, read without looking at a sample column.  The package's Python statement of the VCF
rules (``vcf.read_region``) is the cross-check (``engine="python"``)."""

from __future__ import annotations

import warnings
from typing import Optional, Sequence, Union

import numpy as np

from .genomic_dataclasses import ChromosomeData
from .samples import parse_ind_file
from .vcf import _alleles, read_anc_allele, read_region, read_site_columns


# -- what the reference asks of allel.GenotypeArray, on the plain [sites][individuals][ploidy] array ------


def calls_missing(gt: np.ndarray) -> np.ndarray:
    """[sites][individuals]: the call has a missing allele (GenotypeArray.is_missing)."""
    return np.any(np.asarray(gt) < 0, axis=-1)


def calls_hom_ref(gt: np.ndarray) -> np.ndarray:
    """every allele is the reference allele (GenotypeArray.is_hom_ref)."""
    return np.all(np.asarray(gt) == 0, axis=-1)


def calls_hom_alt(gt: np.ndarray) -> np.ndarray:
    """every allele is the same alternate allele (GenotypeArray.is_hom_alt)."""
    g = np.asarray(gt)
    first = g[..., :1]
    return np.all((first > 0) & (g == first), axis=-1)


# -- sai/utils/utils.py --------------------------------------------------------------------------------


def filter_geno_data(data: ChromosomeData, index: Union[np.ndarray, Sequence[bool]]) -> ChromosomeData:
    """The rows ``index`` keeps (boolean mask or integer indices) of every field (utils.py:189-212)."""
    index = np.asarray(index)
    gt = np.asarray(data.GT)
    rows = lambda column: None if column is None else np.asarray(column)[index]  # noqa: E731 -- blocks of the native ingest carry no REF / ALT
    return ChromosomeData(POS=np.asarray(data.POS)[index], REF=rows(data.REF), ALT=rows(data.ALT),
                          GT=np.compress(index, gt, axis=0) if index.dtype == bool else gt[index])  # fmt: skip


def filter_fixed_variants(data: dict, samples: dict) -> dict:
    """Per population: drop the sites where every sample is homozygous reference, or every sample
    homozygous for an alternate allele (utils.py:359-386)."""
    out = {}
    for population, block in data.items():
        n = len(samples[population])
        fixed = (calls_hom_ref(block.GT).sum(axis=1) == n) | (calls_hom_alt(block.GT).sum(axis=1) == n)
        out[population] = filter_geno_data(block, ~fixed)
    return out


def reshape_genotypes(data: Optional[dict], is_phased: bool) -> None:
    """In place: phased -> haplotype columns [sites][individuals * ploidy]; unphased -> the sum over the
    ploidy axis (utils.py:389-410)."""
    if data is None:
        return
    for block in data.values():
        gt = np.asarray(block.GT)
        n_sites, n_ind, ploidy = gt.shape
        block.GT = gt.reshape(n_sites, n_ind * ploidy) if is_phased else gt.sum(axis=2)


def get_ref_alt_allele(ref, alt, pos) -> tuple[dict, dict]:
    """({pos: REF}, {pos: ALT}) (utils.py:413-432)."""
    return {p: r for p, r in zip(pos, ref)}, {p: a for p, a in zip(pos, alt)}


def flip_snps(data: ChromosomeData, flipped_snps) -> None:
    """In place: every allele call a of the listed positions becomes |a - 1| -- ALT is the ancestral
    allele there (utils.py:540-555; a missing allele, -1, becomes 2, as in the reference)."""
    flipped = np.isin(np.asarray(data.POS), np.asarray(list(flipped_snps), dtype=np.int64))
    gt = np.asarray(data.GT)
    gt[flipped] = np.abs(gt[flipped] - 1)
    data.GT = gt


def check_anc_allele(data: ChromosomeData, anc_allele: dict, c: str) -> ChromosomeData:
    """Keep the sites the ancestral-allele table lists, drop those whose ancestral allele is neither REF
    nor ALT, flip those where it is ALT (utils.py:492-537)."""
    ref_allele, alt_allele = get_ref_alt_allele(data.REF, data.ALT, data.POS)
    table = anc_allele.get(c, {})
    listed = np.intersect1d(np.asarray(list(ref_allele.keys()), dtype=np.int64), np.asarray(list(table.keys()), dtype=np.int64))
    removed, flipped = [], []
    for v in listed.tolist():
        if table[v] not in {ref_allele[v], alt_allele[v]}:
            removed.append(v)
        elif table[v] == alt_allele[v]:
            flipped.append(v)
    data = filter_geno_data(data, np.isin(data.POS, listed))
    if removed:
        data = filter_geno_data(data, ~np.isin(data.POS, removed))
    flip_snps(data, flipped)
    return data


def allele_calls(vcf: str, chr_name: str, samples: Sequence[str], ploidy: int, start=None, end=None, engine: str = "native"):
    """(POS int32 [n], REF [n], ALT [n], calls int8 [n][len(samples)][ploidy]) of one region, unpolarised.

    ``engine="native"``: ``sai_vcf_load`` read at ploidy 1, 2, ..., ``ploidy`` -- it sums a call's first k alleles
    (missing = -1, a call shorter than k padded with -1, longer ones cut: ``numbers={"GT": ploidy}``,
    utils.py:123-138), so allele k = dosage_k - dosage_(k-1), exactly; one multithreaded pass over the file
    per allele.  ``engine="python"``: the readable statement of the same rules, call by call."""
    samples = list(samples)
    if engine == "python":
        records = read_region(vcf, chr_name, samples, start, end)
        cache: dict = {}
        calls = np.empty((len(records), len(samples), ploidy), dtype=np.int8)
        for i, row in enumerate(records.gt):
            for j, call in enumerate(row):
                alleles = cache.get(call)
                if alleles is None:
                    alleles = cache[call] = _alleles(call, ploidy)
                calls[i, j] = alleles
        return records.pos, records.ref, records.alt, calls
    import os

    from .native_vcf import load_dosage

    if not os.path.exists(vcf):
        raise ValueError(f"cannot open VCF {vcf}")
    pos, before, calls = None, None, None
    for k in range(1, ploidy + 1):
        pos_k, dosage, _, _ = load_dosage(vcf, chr_name, samples, [k] * len(samples), start, end, None)
        if calls is None:
            pos, calls = pos_k, np.empty((len(pos_k), len(samples), ploidy), dtype=np.int8)
            before = np.zeros(dosage.shape, dtype=np.int16)
        now = dosage.astype(np.int16)
        calls[:, :, k - 1] = now - before
        before = now
    pos_text, ref, alt = read_site_columns(vcf, chr_name, start, end)
    if not np.array_equal(pos_text, pos):
        raise ValueError(f"{vcf}: the tokenizer and the fixed-column reader disagree about the records of {chr_name}")
    return pos, ref, alt, calls


def read_geno_data(vcf: str, ind_samples: dict, chr_name: str, ploidy: int = 2, start: int = None, end: int = None,
                   anc_allele_file: Optional[str] = None, filter_missing: bool = True, engine: str = "native") -> Optional[dict]:  # fmt: skip
    """{population: ChromosomeData} of one chromosome (region) with per-allele calls (utils.py:78-186):
    first ALT allele only (``alt_number=1``), calls padded with -1 / cut to ``ploidy``
    (``numbers={"GT": ploidy}``); ``filter_missing`` drops the sites where a sample of the population has
    a missing allele; with an ancestral-allele file the block is polarised (``check_anc_allele``).  None
    when the region holds no record."""
    chr_name = str(chr_name)
    region = f"{chr_name}" if start is None and end is None else f"{chr_name}:{start}-{end}"
    all_samples = [s for names in ind_samples.values() for s in names]
    try:
        pos, ref_list, alt_list, gt = allele_calls(vcf, chr_name, all_samples, ploidy, start, end, engine)
    except Exception as e:  # noqa: BLE001 -- utils.py:139-140
        raise ValueError(f"Failed to read VCF file {vcf} from {region}: {e}") from e
    if len(pos) == 0:
        return None
    ref, alt = np.array(ref_list, dtype=object).astype(str), np.array(alt_list, dtype=object).astype(str)
    anc_alleles = read_anc_allele(anc_allele_file, chr_name, start, end) if anc_allele_file else None
    out, column = {}, {s: j for j, s in enumerate(all_samples)}
    for population, names in ind_samples.items():
        block = ChromosomeData(POS=pos.copy(), REF=ref.copy(), ALT=alt.copy(), GT=gt[:, [column[s] for s in names]])
        missing = calls_missing(block.GT).sum(axis=1) != 0
        if filter_missing and missing.any():
            block = filter_geno_data(block, ~missing)
        if anc_alleles:
            block = check_anc_allele(block, anc_alleles, chr_name)
        out[population] = block
    return out


def load_population_data(vcf_file, chr_name, sample_file, anc_allele_file, start, end, is_phased, filter_flag, filter_missing,
                         ploidy_config, group, engine: str = "native"):  # fmt: skip
    """(data, samples) of one group (utils.py:649-761): every population of the sample file that has a
    ploidy entry is read at ITS ploidy, optionally stripped of its fixed variants, then reshaped."""
    if sample_file is None:
        return None, None
    samples = parse_ind_file(sample_file)
    if group not in ploidy_config.root:
        raise ValueError(f"Ploidy configuration missing group '{group}'.")
    group_ploidies = ploidy_config.root[group]
    for population in group_ploidies:
        if population not in samples:
            raise ValueError(f"Population '{population}' in ploidy_config[{group}] not found in sample file: {sample_file}")
    data = {}
    for population, names in samples.items():
        if population not in group_ploidies:
            warnings.warn(f"Population '{population}' found in sample file but not in ploidy_config[{group}]; skipping.", RuntimeWarning)
            continue
        try:
            blocks = read_geno_data(vcf=vcf_file, ind_samples={population: names}, chr_name=chr_name, start=start, end=end,
                                    anc_allele_file=anc_allele_file, filter_missing=filter_missing, ploidy=group_ploidies[population],
                                    engine=engine)  # fmt: skip
        except Exception as e:  # noqa: BLE001 -- utils.py:735-738
            raise ValueError(f"Failed to read VCF data for {sample_file}, population '{population}': {e}")
        if blocks is None:
            continue
        if filter_flag:
            blocks = filter_fixed_variants(blocks, {population: names})
        reshape_genotypes(blocks, is_phased)
        data[population] = blocks[population]
    return (data if data else None), samples
