"""Load the ref / tgt / src populations of one chromosome region (mirror of
sai/utils/utils.py:215-356 ``read_data`` and :649-761 ``_load_population_data`` as used by
WindowGenerator: unphased, no fixed-variant or missing-data filtering)."""

from __future__ import annotations

import warnings
from typing import Optional

from .genomic_dataclasses import ChromosomeData
from .samples import parse_ind_file
from .vcf import dosage_matrices, polarisation_masks, read_anc_allele, read_region


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
) -> dict[str, tuple[Optional[dict[str, ChromosomeData]], Optional[dict[str, list[str]]]]]:
    """{"ref": (data, samples), "tgt": ..., "src": ..., "outgroup": (None, None)}.

    ``data`` maps population -> ChromosomeData for the populations that have a ploidy entry
    (others are skipped with the reference's RuntimeWarning, utils.py:722-728); a population
    with a ploidy but no samples is a ValueError (:713-718); ``data`` is None when the region
    holds no record.  The VCF is parsed once for all groups (the reference re-reads it per
    population); outgroups are outside this path."""
    chr_name = str(chr_name)
    groups = [("ref", ref_ind_file), ("tgt", tgt_ind_file), ("src", src_ind_file)]
    samples_by_group: dict[str, Optional[dict[str, list[str]]]] = {}
    wanted: list[str] = []
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
                wanted.extend(n for n in names if n not in wanted)

    results: dict = {"outgroup": (None, None)}
    if not wanted:
        for group, _ in groups:
            results[group] = (None, samples_by_group[group])
        return results

    try:
        region = read_region(vcf_file, chr_name, wanted, start, end)
    except Exception as e:  # utils.py:139-140
        where = chr_name if start is None and end is None else f"{chr_name}:{start}-{end}"
        raise ValueError(f"Failed to read VCF file {vcf_file} from {where}: {e}") from e
    column = {name: i for i, name in enumerate(region.samples)}

    keep = flip = None
    if anc_allele_file and len(region):
        anc = read_anc_allele(anc_allele_file, chr_name, start, end)
        keep, flip = polarisation_masks(region, anc.get(chr_name, {}))

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
            if len(region) == 0:
                continue
            ploidy = ploidy_config.root[group][population]
            dos, fdos = dosage_matrices(region, [column[n] for n in names], ploidy)
            pos, ref, alt = region.pos, region.ref, region.alt
            if keep is not None:
                dos[flip] = fdos[flip]
                dos = dos[keep]
                pos = pos[keep]
                ref = [r for r, k in zip(ref, keep) if k]
                alt = [a for a, k in zip(alt, keep) if k]
            data[population] = ChromosomeData(POS=pos.copy(), REF=list(ref), ALT=list(alt), GT=dos)
        results[group] = (data if data else None, samples)
    return results
