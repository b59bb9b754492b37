"""Load the ref / tgt / src populations of one chromosome region (mirror of sai/utils/utils.py:215-356
``read_data`` and :649-761 ``_load_population_data``).  ``read_data`` has the reference's signature and
defaults; the combination WindowGenerator asks for (window_generator.py:105-120: unphased, no
fixed-variant or missing-call filter) is served by the native tokenizers as int8 dosages
(``read_dosage_data`` on the host, ``read_data_device`` in HBM), every other combination by the
allele-level reader of ``geno.py`` (calls from the same native tokenizer, allele by allele)."""

from __future__ import annotations

import os
import warnings
from typing import Optional

import numpy as np

from .genomic_dataclasses import ChromosomeData
from .samples import parse_ind_file


def read_data(
    vcf_file: str,
    chr_name: str,
    ploidy_config,
    ref_ind_file: Optional[str],
    tgt_ind_file: Optional[str],
    src_ind_file: Optional[str],
    out_ind_file: Optional[str] = None,
    anc_allele_file: Optional[str] = None,
    start: int = None,
    end: int = None,
    is_phased: bool = True,
    filter_ref: bool = True,
    filter_tgt: bool = True,
    filter_src: bool = False,
    filter_out: bool = False,
    filter_missing: bool = True,
    engine: str = "native",
) -> dict[str, tuple[Optional[dict[str, ChromosomeData]], Optional[dict[str, list[str]]]]]:
    """{"ref": (data, samples), "tgt": ..., "src": ..., "outgroup": ...} with the reference's options
    (utils.py:215-356): ``is_phased`` = haplotype columns [sites][individuals * ploidy] instead of the
    dosage [sites][individuals]; ``filter_<group>`` = drop the variants fixed in a population;
    ``filter_missing`` = drop the sites where a sample of the population has a missing allele.  Blocks
    carry REF / ALT.  All options off and unphased is the ``score`` path's request: it goes to the native
    tokenizer (``read_dosage_data``: int8 dosages, REF / ALT not kept)."""
    if not (is_phased or filter_ref or filter_tgt or filter_src or filter_out or filter_missing):
        return read_dosage_data(vcf_file, chr_name, ploidy_config, ref_ind_file, tgt_ind_file, src_ind_file, out_ind_file,
                                anc_allele_file, start, end, engine)  # fmt: skip
    from .geno import load_population_data

    results = {}
    for group, ind_file, fixed in (("ref", ref_ind_file, filter_ref), ("tgt", tgt_ind_file, filter_tgt),
                                   ("src", src_ind_file, filter_src), ("outgroup", out_ind_file, filter_out)):  # fmt: skip
        if ind_file is None or (group == "outgroup" and group not in ploidy_config.root):  # utils.py:331-337
            results[group] = (None, None)
            continue
        results[group] = load_population_data(vcf_file, str(chr_name), ind_file, anc_allele_file, start, end, is_phased, fixed,
                                              filter_missing, ploidy_config, group, engine)  # fmt: skip
    return results


def read_dosage_data(
    vcf_file: str,
    chr_name: str,
    ploidy_config,
    ref_ind_file: Optional[str],
    tgt_ind_file: Optional[str],
    src_ind_file: Optional[str],
    out_ind_file: Optional[str] = None,
    anc_allele_file: Optional[str] = None,
    start: int = None,
    end: int = None,
    engine: str = "native",
) -> dict[str, tuple[Optional[dict[str, ChromosomeData]], Optional[dict[str, list[str]]]]]:
    """``read_data(..., is_phased=False, filter_*=False, filter_missing=False)``:
    {"ref": (data, samples), "tgt": ..., "src": ..., "outgroup": (data, samples) or (None, None)}.

    ``data`` maps population -> ChromosomeData for the populations that have a ploidy entry
    (others are skipped with the reference's RuntimeWarning, utils.py:722-728); a population
    with a ploidy but no samples is a ValueError (:713-718); ``data`` is None when the region
    holds no record.  The VCF is parsed once per distinct ploidy for all groups (the reference
    re-reads it per population) by the native tokenizer of libsaihip (``engine="native"``) or by
    the Python statement of the same rules (``engine="python"``, used by the tests as the
    cross-check); REF/ALT are not kept (nothing reads them after polarisation)."""
    chr_name = str(chr_name)
    groups = [("ref", ref_ind_file), ("tgt", tgt_ind_file), ("src", src_ind_file)]
    # utils.py:331-337: an outgroup file only counts when the ploidy section names outgroups
    if out_ind_file is not None and "outgroup" in ploidy_config.root:
        groups.append(("outgroup", out_ind_file))
    samples_by_group: dict[str, Optional[dict[str, list[str]]]] = {}
    wanted: set[str] = set()
    for group, ind_file in groups:
        if ind_file is None:
            samples_by_group[group] = None
            continue
        samples = parse_ind_file(ind_file)
        if group not in ploidy_config.root:
            raise ValueError(f"Ploidy configuration missing group '{group}'.")
        for population in ploidy_config.root[group]:
            if population not in samples:
                raise ValueError(
                    f"Population '{population}' in ploidy_config[{group}] not found in sample file: {ind_file}"
                )
        samples_by_group[group] = samples
        for population, names in samples.items():
            if population in ploidy_config.root[group]:
                wanted.update(names)

    results: dict = {"outgroup": (None, None)}
    if not wanted:
        for group, _ in groups:
            results[group] = (None, samples_by_group[group])
        return results

    loader = _load_python if engine == "python" else _load_native
    where = chr_name if start is None and end is None else f"{chr_name}:{start}-{end}"
    # one pass per distinct ploidy (normally one or two), each over the samples that need it
    by_ploidy: dict[int, list[str]] = {}
    seen: dict[int, set[str]] = {}
    for group, _ in groups:
        samples = samples_by_group[group]
        if samples is None:
            continue
        for population, names in samples.items():
            if population in ploidy_config.root[group]:
                ploidy = ploidy_config.root[group][population]
                bucket, have = by_ploidy.setdefault(ploidy, []), seen.setdefault(ploidy, set())
                for n in names:  # first occurrence keeps its place (a sample may sit in several populations)
                    if n not in have:
                        have.add(n)
                        bucket.append(n)
    loaded = {}
    for ploidy, names in by_ploidy.items():
        try:
            pos, dos, n_matched, n_anc = loader(vcf_file, chr_name, names, ploidy, start, end, anc_allele_file)
        except FileNotFoundError:
            raise
        except Exception as e:  # utils.py:139-140
            raise ValueError(f"Failed to read VCF file {vcf_file} from {where}: {e}") from e
        if anc_allele_file and n_matched and n_anc == 0:  # read_anc_allele, utils.py:480-487
            if start is not None or end is not None:
                raise ValueError(f"No ancestral allele is found for chromosome {chr_name} in the region {start}-{end}.")
            raise ValueError(f"No ancestral allele is found for chromosome {chr_name}.")
        loaded[ploidy] = (pos, dos, {n: i for i, n in enumerate(names)}, n_matched)

    for group, _ in groups:
        samples = samples_by_group[group]
        if samples is None:
            results[group] = (None, None)
            continue
        data: dict[str, ChromosomeData] = {}
        for population, names in samples.items():
            if population not in ploidy_config.root[group]:
                warnings.warn(
                    f"Population '{population}' found in sample file but not in ploidy_config[{group}]; skipping.",
                    RuntimeWarning,
                )
                continue
            pos, dos, column, n_matched = loaded[ploidy_config.root[group][population]]
            if n_matched == 0:  # no record in the region: the reference's "vcf_data is None" case
                continue
            cols = [column[n] for n in names]
            if cols and cols == list(range(cols[0], cols[0] + len(cols))):  # the usual case: one block of columns
                gt = np.ascontiguousarray(dos[:, cols[0] : cols[0] + len(cols)])
            else:
                gt = np.ascontiguousarray(dos[:, cols])
            data[population] = ChromosomeData(POS=pos.copy(), REF=None, ALT=None, GT=gt)
        results[group] = (data if data else None, samples)
    return results


def read_data_device(eng, vcf_file: str, chr_name: str, ploidy_config, ref_ind_file, tgt_ind_file, src_ind_file,
                     out_ind_file=None, anc_allele_file=None, start: int = None, end: int = None):
    """``read_data`` with the genotypes left in HBM: one streaming pass over the file (text over PCIe,
    tokenised on the GPU, ``device_vcf.load_dosage_device``) for all populations and ploidies, then one
    re-tiling launch per population straight from the shared [record][sample] block.  Returns
    ``(results, pos_dev)``: ``results`` as ``read_data`` gives it, except that every ``GT`` is a
    ``TiledPop``; ``pos_dev`` = the int32 device copy of the positions (None without data)."""
    import torch

    from .device_vcf import load_dosage_device

    chr_name = str(chr_name)
    groups = [("ref", ref_ind_file), ("tgt", tgt_ind_file), ("src", src_ind_file)]
    if out_ind_file is not None and "outgroup" in ploidy_config.root:
        groups.append(("outgroup", out_ind_file))
    # One streaming pass maps a VCF column to ONE output slot, so a sample that sits in populations of
    # different ploidy (the reference reads every population on its own, with its own ploidy:
    # utils.py:123-138) is tokenised in a further pass: passes[k] holds each sample at most once.
    samples_by_group, column, passes = {}, {}, []
    for group, ind_file in groups:
        if ind_file is None:
            samples_by_group[group] = None
            continue
        samples = parse_ind_file(ind_file)
        if group not in ploidy_config.root:
            raise ValueError(f"Ploidy configuration missing group '{group}'.")
        for population in ploidy_config.root[group]:
            if population not in samples:
                raise ValueError(
                    f"Population '{population}' in ploidy_config[{group}] not found in sample file: {ind_file}"
                )
        samples_by_group[group] = samples
        for population, pop_names in samples.items():
            if population not in ploidy_config.root[group]:
                continue
            ploidy = ploidy_config.root[group][population]
            for nme in pop_names:  # one output column per (sample, ploidy): the first occurrence keeps its place
                if (nme, ploidy) not in column:
                    k = next((i for i, p in enumerate(passes) if nme not in p["seen"]), len(passes))
                    if k == len(passes):
                        passes.append({"names": [], "ploidies": [], "seen": set()})
                    column[(nme, ploidy)] = (k, len(passes[k]["names"]))
                    passes[k]["names"].append(nme)
                    passes[k]["ploidies"].append(ploidy)
                    passes[k]["seen"].add(nme)
    results: dict = {"outgroup": (None, None)}
    if not passes:
        for group, _ in groups:
            results[group] = (None, samples_by_group[group])
        return results, None
    where = chr_name if start is None and end is None else f"{chr_name}:{start}-{end}"
    if not os.path.exists(vcf_file):
        raise ValueError(f"Failed to read VCF file {vcf_file} from {where}: cannot open VCF {vcf_file}")
    try:
        blocks = []
        for p in passes:  # one pass unless a sample is read at two ploidies
            pos, dos, n_matched, n_anc = load_dosage_device(eng, vcf_file, chr_name, p["names"], p["ploidies"], start, end,
                                                            anc_allele_file)  # fmt: skip
            blocks.append(dos)
    except FileNotFoundError:
        raise
    except Exception as e:  # utils.py:139-140
        raise ValueError(f"Failed to read VCF file {vcf_file} from {where}: {e}") from e
    if anc_allele_file and n_matched and n_anc == 0:  # read_anc_allele, utils.py:480-487
        if start is not None or end is not None:
            raise ValueError(f"No ancestral allele is found for chromosome {chr_name} in the region {start}-{end}.")
        raise ValueError(f"No ancestral allele is found for chromosome {chr_name}.")
    pos_dev = torch.from_numpy(pos).to(eng.device) if n_matched else None
    for group, _ in groups:
        samples = samples_by_group[group]
        if samples is None:
            results[group] = (None, None)
            continue
        data: dict[str, ChromosomeData] = {}
        for population, pop_names in samples.items():
            if population not in ploidy_config.root[group]:
                warnings.warn(
                    f"Population '{population}' found in sample file but not in ploidy_config[{group}]; skipping.",
                    RuntimeWarning,
                )
                continue
            if n_matched == 0:
                continue
            ploidy = ploidy_config.root[group][population]
            where_cols = [column[(nme, ploidy)] for nme in pop_names]
            used = {k for k, _ in where_cols}
            if len(used) == 1:
                tiled = eng.tile_columns(blocks[used.pop()], [c for _, c in where_cols])
            else:  # the population's samples were tokenised in different passes: gather its columns first
                tiled = eng._tile_device(torch.stack([blocks[k][:, c] for k, c in where_cols], dim=1).contiguous())
            data[population] = ChromosomeData(POS=pos, REF=None, ALT=None, GT=tiled)
        results[group] = (data if data else None, samples)
    return results, pos_dev


def _load_native(vcf_file, chr_name, names, ploidy, start, end, anc_allele_file):
    """libsaihip's multithreaded tokenizer (sai_amd/csrc/vcf_ingest.cpp)."""
    from .native_vcf import load_dosage

    if not os.path.exists(vcf_file):
        raise ValueError(f"cannot open VCF {vcf_file}")
    return load_dosage(vcf_file, chr_name, names, [ploidy] * len(names), start, end, anc_allele_file)


def _load_python(vcf_file, chr_name, names, ploidy, start, end, anc_allele_file):
    """The readable statement of the same rules (sai_amd/utils/vcf.py); tests compare the two."""
    from .vcf import dosage_matrices, polarisation_masks, read_anc_allele, read_region

    region = read_region(vcf_file, chr_name, names, start, end)
    dos, fdos = dosage_matrices(region, list(range(len(names))), ploidy)
    pos, n_matched, n_anc = region.pos, len(region), 0
    if anc_allele_file and len(region):
        try:
            anc = read_anc_allele(anc_allele_file, chr_name, start, end)
        except ValueError:
            anc = {}
        table = anc.get(chr_name, {})
        n_anc = len(table)
        keep, flip = polarisation_masks(region, table)
        dos[flip] = fdos[flip]
        dos, pos = dos[keep], pos[keep]
    return pos, dos, n_matched, n_anc
