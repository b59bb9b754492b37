"""``score`` and ``outlier`` -- the functions behind ``sai score`` / ``sai outlier`` (mirror of
sai/sai.py:33-230)."""

from __future__ import annotations

import os
import warnings
from pathlib import Path

import yaml

from .configs import GlobalConfig
from .generators import ChunkGenerator
from .preprocessors import ChunkPreprocessor
from .utils import UniqueKeyLoader

_POLARISED = ("fd", "df", "Danc", "Dplus")


def load_config(config: str) -> GlobalConfig:
    """YAML -> GlobalConfig with the reference's error behaviour (sai.py:65-73)."""
    try:
        with open(config, "r") as f:
            config_dict = yaml.load(f, Loader=UniqueKeyLoader)
    except FileNotFoundError:
        raise FileNotFoundError(f"Configuration file '{config}' not found.")
    except yaml.YAMLError as e:
        raise ValueError(f"Error parsing YAML configuration file '{config}': {e}")
    return GlobalConfig(**config_dict)


def header_line(stat_config, ploidy_config) -> str:
    """sai.py:107-131: fixed columns + one column per statistic in YAML order (U/Q always a
    single column; a statistic set to False is left out)."""
    cols = ["Chrom", "Start", "End", "Ref", "Tgt", "Src", "Outgroup", "N(Variants)"]
    src_pops = list(ploidy_config.root["src"].keys())
    for name, value in stat_config.root.items():
        if name not in ("U", "Q") and value is False:
            continue
        if name in ("U", "Q") or len(src_pops) <= 1:
            cols.append(name)
        else:
            cols.extend(f"{name}.{sp}" for sp in src_pops)
    return "\t".join(cols) + "\n"


def write_headers(output_file: str, stat_config, ploidy_config) -> None:
    """Create the output directory, the TSV with its header and the .U.log/.Q.log files
    (sai.py:133-144)."""
    directory = os.path.dirname(output_file)
    if directory:
        os.makedirs(directory, exist_ok=True)
    with open(output_file, "w") as f:
        f.write(header_line(stat_config, ploidy_config))
    for key in ("U", "Q"):
        if key in stat_config.root:
            with open(Path(output_file).with_suffix(f".{key}.log"), "w") as f:
                f.write(f"Chrom\tStart\tEnd\t{key}_SNP\n")


def require_polarised_input(stat_config, anc_allele_file) -> None:
    """fd, df, Danc and Dplus need polarised data (sai.py:79-84)."""
    if anc_allele_file is not None:
        return
    wanted = [name for name in stat_config.root if name in _POLARISED]
    if wanted:
        raise ValueError(
            f"The {wanted[0]} statistic requires polarized data, please provide the ancestral allele information with `--anc-alleles`."
        )


def chunk_preprocessor_for(cfg: GlobalConfig, vcf_file, win_len, win_step, output_file, anc_allele_file) -> ChunkPreprocessor:
    """The chunk driver of a run: the sample lists come from the configuration's ``populations``
    section (sai.py:95-107).  Shared by ``score`` and ``sai_amd.distributed.score_sharded``."""
    files = {group: cfg.populations.get_population(group) for group in ("ref", "tgt", "src", "outgroup")}
    return ChunkPreprocessor(vcf_file, files["ref"], files["tgt"], files["src"], files["outgroup"], win_len, win_step,
                             output_file, cfg.ploidies, cfg.statistics, anc_allele_file=anc_allele_file)  # fmt: skip


def _reads_in_one_pass(vcf_file: str) -> bool:
    """A plain-text VCF on a machine with a GPU: the chromosome's first and last position (a host scan of
    the file, chunk_generator.py:64-82) are found WHILE the one streaming read fills HBM -- the scan was 10
    of the 24 ms of a 482 MB file.  Compressed files keep the order scan, then read: an indexed one scans
    two records, and the scan of an unindexed bgzip file is itself a pass of the GPU reader."""
    if str(vcf_file).endswith((".gz", ".bgz")) or os.environ.get("SAI_AMD_INGEST", "device") == "host":
        return False
    if os.environ.get("SAI_AMD_ONE_PASS", "1") == "0" or not os.path.isfile(vcf_file):
        return False
    try:
        import torch

        return bool(torch.cuda.is_available())
    except ImportError:
        return False


def _is_device_failure(exc: BaseException) -> bool:
    """A HIP call, the device or its memory failed -- anywhere in the exception's chain (the readers wrap what
    they meet in the reference's ``ValueError("Failed to read VCF file ...")``, utils.py:139-140)."""
    from . import _ffi

    seen = set()
    while exc is not None and id(exc) not in seen:
        seen.add(id(exc))
        if isinstance(exc, _ffi.SaiHipError) and exc.status in (_ffi.SAI_ERR_HIP, _ffi.SAI_ERR_NO_DEVICE):
            return True
        if isinstance(exc, MemoryError) or type(exc).__name__ in ("OutOfMemoryError", "AcceleratorError"):
            return True
        if isinstance(exc, RuntimeError) and not isinstance(exc, _ffi.SaiHipError) and ("HIP error" in str(exc) or "hip" in str(exc)[:200].lower()):
            return True
        exc = exc.__cause__ or exc.__context__
    return False


def _scan_while_reading(driver: ChunkPreprocessor, vcf_file: str, chr_name: str):
    """((first, last) of the scan, what ``ChunkPreprocessor.preload`` read meanwhile or None).  The scan's
    answer decides: a chromosome it does not find is reported as ``ChunkGenerator`` reports it, and a
    read that failed over the FILE (a malformed line, an unknown sample, no ancestral allele ...) is left to be
    repeated -- and to fail with its own message -- in the usual order.  A failure of the device is not: it is
    raised here, by the call that met it, not after a second full read."""
    from concurrent.futures import ThreadPoolExecutor

    from .utils.native_vcf import scan_first_last

    with ThreadPoolExecutor(1) as pool:
        scan = pool.submit(scan_first_last, vcf_file, chr_name)  # host threads inside libsaihip; the GIL is released
        try:
            preloaded = driver.preload(chr_name)
        except Exception as exc:  # noqa: BLE001 -- a file-level error: repeated by run_compact without the preload
            if _is_device_failure(exc):
                scan.result()
                raise
            preloaded = None
        span = scan.result()
    return span, preloaded


def chunks_for_memory(vcf_file: str) -> int:
    """How many ChunkGenerator chunks a one-process `score` cuts the chromosome into so that a chunk's genotypes
    fit the GPU: 1 (the reference's one-process form, sai.py:86-93 with one worker) unless the file promises more
    int8 genotype bytes than the budget -- then ceil(bytes / budget), the grain the sharded route already uses
    (chunk_generator.py:111-142), chunk after chunk through the same GPU, the output files byte-identical.
    The estimate is the file's: a genotype is about four bytes of VCF text ("0|1" + tab; bgzip shrinks genotype
    text about twelvefold), of which one int8 dosage stays resident -- besides the reader's staging and the tiled
    copy per population, hence a budget of a quarter of the free HBM.  ``SAI_AMD_HBM_BUDGET_BYTES`` overrides it."""
    try:
        size = os.path.getsize(vcf_file)
    except OSError:
        return 1
    resident = size * 3 if str(vcf_file).endswith((".gz", ".bgz")) else size // 4
    raw = os.environ.get("SAI_AMD_HBM_BUDGET_BYTES", "")
    if raw:
        budget = int(raw)
        if budget < 1:
            raise ValueError("SAI_AMD_HBM_BUDGET_BYTES must be a positive integer")
    else:
        try:
            import torch

            if not torch.cuda.is_available():
                return 1
            budget = torch.cuda.mem_get_info()[0] // 4
        except (ImportError, RuntimeError):
            return 1
    return max(1, -(-resident // max(budget, 1)))


def _score_over_ranks(vcf_file, chr_name, win_len, win_step, anc_allele_file, output_file, config) -> None:
    """This process is one rank of a launched job: take the sharded route and leave the group cleanly."""
    from .distributed import score_sharded, shutdown_process_group
    from .launcher import chunks_per_worker

    try:
        score_sharded(vcf_file, chr_name, win_len, win_step, anc_allele_file, output_file, config,
                      chunks_per_rank=chunks_per_worker())  # fmt: skip
    finally:
        shutdown_process_group()


def _score_cli_arguments(vcf_file, chr_name, win_len, win_step, anc_allele_file, output_file, config, num_workers) -> list:
    argv = ["score", "--vcf", vcf_file, "--chr-name", chr_name, "--win-len", win_len, "--win-step", win_step,
            "--output", output_file, "--config", config, "--num-workers", num_workers]  # fmt: skip
    if anc_allele_file is not None:
        argv += ["--anc-alleles", anc_allele_file]
    return [str(a) for a in argv]


def score(vcf_file: str, chr_name: str, win_len: int, win_step: int, anc_allele_file: str, output_file: str, config: str,
          num_workers: int) -> None:  # fmt: skip
    """Sliding-window scores of one chromosome, written as the reference writes them (TSV +
    ``.U.log`` + ``.Q.log``; interface of sai.py:33-42).

    ``num_workers`` is the reference's worker-process count (sai.py:42, grain ``num_workers * 8`` chunks at
    :91) with one process per GPU: 1 computes every window in this process; N > 1 starts N ranks as a
    child job (``sai_amd.launcher``: before this process has touched the GPU) whose ranks cut the window
    list into ``N * 8`` ChunkGenerator chunks, each rank reading and scoring its own contiguous share on
    its own GPU, with one final gather to rank 0, which writes the files -- byte-identical to the
    one-process files for any N (``sai_amd.distributed.score_sharded``).  Inside such a job
    (``WORLD_SIZE`` > 1: torchrun's environment) the call IS a rank and takes the sharded route."""
    from . import launcher

    num_workers = 1 if num_workers is None else int(num_workers)
    if num_workers < 1:
        raise ValueError("`num_workers` must be a positive integer.")
    if launcher.in_rank_job():
        _score_over_ranks(vcf_file, chr_name, win_len, win_step, anc_allele_file, output_file, config)
        return
    if num_workers > 1:
        # the errors a one-process run raises before any work (configuration, polarisation) are raised here,
        # by the caller's own process, not as a failed child job
        cfg = load_config(config)
        require_polarised_input(cfg.statistics, anc_allele_file)
        rc = launcher.launch_ranks(
            num_workers, _score_cli_arguments(vcf_file, chr_name, win_len, win_step, anc_allele_file, output_file, config, num_workers),
            module="sai_amd", who="sai score")  # fmt: skip
        if rc != 0:
            raise launcher.RankJobFailed(num_workers, rc)
        return
    cfg = load_config(config)
    require_polarised_input(cfg.statistics, anc_allele_file)
    driver = chunk_preprocessor_for(cfg, vcf_file, win_len, win_step, output_file, anc_allele_file)
    span, preloaded = None, None
    n_chunks = chunks_for_memory(vcf_file)  # 1 unless the chromosome's genotypes would not fit the GPU at once
    if n_chunks == 1 and _reads_in_one_pass(vcf_file):
        span, preloaded = _scan_while_reading(driver, vcf_file, str(chr_name))
    chunks = ChunkGenerator(vcf_file=vcf_file, chr_name=chr_name, window_size=win_len, step_size=win_step,
                            num_chunks=n_chunks, span=span)  # fmt: skip
    write_headers(output_file, cfg.statistics, cfg.ploidies)
    # numeric batches -> text, natively; the item-dictionary route (driver.run + process_items)
    # writes the same bytes and stays what plug-ins and the sharded executors use
    tasks = list(chunks.get())
    if len(tasks) == 1:  # the usual run: the rows of the first windows are written while the GPU scores the later ones
        chunk = tasks[0]
        fits = preloaded is not None and (preloaded[2] is None or (chunk["start"] <= preloaded[2][0] and preloaded[2][1] <= chunk["end"]))
        driver.run_and_write(**chunk, preloaded=preloaded[:2] if fits else None)
    else:  # chunk after chunk through the GPU; the files are combination-major, so the rows wait for the last chunk
        driver.write_results([driver.run_compact(**chunk) for chunk in tasks])
    if os.environ.get("SAI_AMD_KEEP_INGEST_BUFFERS", "1") == "0":
        # the readers' staging (about 650 MB of HBM + 125 MB pinned for a large bgzip file) is kept for the
        # next call by default -- allocating and page-locking it costs more than a small `score`
        from .engine import Engine

        Engine.get().release_ingest_buffers()


def outlier(score_file: str, output_prefix: str, quantile: float) -> None:
    """Per statistic column of a score table, write the rows beyond the ``quantile`` of that
    column to ``{output_prefix}.{column}.{quantile}.outliers.tsv`` (sai.py:154-230): columns
    after ``N(Variants)`` are statistics; ``U*`` columns keep rows strictly above the threshold,
    the others rows at or above it; a column without numbers, or with a single distinct value,
    gives a header-only file and a UserWarning; rows are naturally sorted by Chrom/Start/End.
    Host-side post-processing of a small table (pandas, like the reference); no GPU involved."""
    import pandas as pd

    from .utils import natsorted_df

    df = pd.read_csv(score_file, sep="\t", na_values=["nan"], index_col=False)
    cols = list(df.columns)
    if "N(Variants)" in cols:
        metric_cols = cols[cols.index("N(Variants)") + 1 :]
    else:  # sai.py:188-194
        skip = {"Chrom", "Start", "End", "Ref", "Tgt", "Src"}
        metric_cols = [c for c in cols if c not in skip and pd.to_numeric(df[c], errors="coerce").notna().any()]
    if not metric_cols:
        raise ValueError("No metric columns found.")
    for col in metric_cols:
        numeric = pd.to_numeric(df[col], errors="coerce")
        present = numeric.dropna()
        if present.empty:
            warnings.warn(f"Column '{col}' has no numeric values; writing empty result.", UserWarning)
            out = pd.DataFrame(columns=df.columns)
        elif present.nunique() == 1:
            warnings.warn(
                f"Column '{col}' has only one unique value ({present.iloc[0]}); writing empty result.", UserWarning
            )
            out = pd.DataFrame(columns=df.columns)
        else:
            thr = present.quantile(quantile)
            out = df[numeric > thr] if col.startswith("U") else df[numeric >= thr]
            if not out.empty:
                out = natsorted_df(out.reset_index(drop=True))
        out.astype(str).to_csv(f"{output_prefix}.{col}.{quantile}.outliers.tsv", index=False, sep="\t")
