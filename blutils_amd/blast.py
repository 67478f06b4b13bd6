"""`blu blastn run-with-consensus`: BLAST fan-out + consensus.  Host orchestration only — BLAST stays an external
process (last row).  Mirrors, with the same names and arguments:

    BlastBuilder                        core/src/domain/dtos/blast_builder.rs:58-127
    FileOrStdin::sequence_content       core/src/domain/dtos/file_or_stdin.rs:174-215
    validate_blast_database             core/src/use_cases/shared/validate_blast_database.rs:5-60
    ExecuteBlastnProcRepository.run     adapters/proc/src/execute_blast.rs:11-57
    run_parallel_blast                  core/src/use_cases/run_blast_and_build_consensus/run_parallel_blast.rs:35-168
    run_blast_and_build_consensus       core/src/use_cases/run_blast_and_build_consensus/mod.rs:22-72

The consensus step is the GPU pipeline (blutils_amd.pipeline); the config is written into the result document the way
write_blutils_output does (run id on every result, `subjectReads` reduced to its file name)."""
from __future__ import annotations

import glob
import json
import os
import subprocess
import sys
import uuid
from concurrent.futures import ThreadPoolExecutor
from dataclasses import dataclass, field, replace
from typing import List, Optional, Sequence as Seq

import numpy as np

from . import pipeline

BLUTILS_VERSION = "8.3.1"        # the reference version this engine mirrors (env!("CARGO_PKG_VERSION"))
OUT_FORMAT_6 = "6 qseqid saccver staxid pident length mismatch gapopen qstart qend sstart send evalue bitscore"
CHUNK_SIZE = 50                  # run_parallel_blast.rs:100


class BlastError(Exception):
    pass


def _shortest_f32(x: float):
    """(digits, exp10): the shortest decimal digits that read back to the same f32, value = digits x 10^exp10."""
    r = np.format_float_scientific(np.float32(x), unique=True, trim="-")      # e.g. '1.e-05' -> '1e-05'
    mant, e = r.split("e")
    sign = "-" if mant.startswith("-") else ""
    mant = mant.lstrip("-")
    ip, _, fp = mant.partition(".")
    digits = (ip + fp).lstrip("0") or "0"
    exp10 = int(e) - len(fp)
    if digits == "0":
        return sign + "0", 0
    stripped = digits.rstrip("0")
    exp10 += len(digits) - len(stripped)
    return sign + stripped, exp10


def _plain(digits: str, exp10: int) -> str:
    sign = "-" if digits.startswith("-") else ""
    d = digits.lstrip("-")
    if exp10 >= 0:
        return sign + d + "0" * exp10
    if len(d) > -exp10:
        return sign + d[:exp10] + "." + d[exp10:]
    return sign + "0." + "0" * (-exp10 - len(d)) + d


@dataclass
class BlastBuilder:
    subject_reads: str
    taxon: str
    is_config: bool = True
    run_id: str = field(default_factory=lambda: str(uuid.uuid4()))
    blutils_version: str = BLUTILS_VERSION
    out_format: str = OUT_FORMAT_6
    max_target_seqs: int = 10
    perc_identity: int = 80
    query_cov: int = 80
    strand: str = "both"
    e_value: float = 0.001       # f32 in the reference
    word_size: int = 15

    @classmethod
    def default(cls, subject_reads: str, taxon: str) -> "BlastBuilder":
        return cls(subject_reads=subject_reads, taxon=taxon)

    def with_max_target_seqs(self, v): return replace(self, max_target_seqs=int(v))
    def with_perc_identity(self, v): return replace(self, perc_identity=int(v))
    def with_query_cov(self, v): return replace(self, query_cov=int(v))
    def with_strand(self, v): return replace(self, strand=str(v))
    def with_e_value(self, v): return replace(self, e_value=float(v))
    def with_word_size(self, v): return replace(self, word_size=int(v))

    def e_value_text(self) -> str:
        """`e_value.to_string()` (Display for f32: shortest round-trip digits, never an exponent) — the `-evalue` argument."""
        digits, exp10 = _shortest_f32(self.e_value)
        return _plain(digits, exp10)

    def e_value_json(self) -> str:
        """serde_json / serde_yaml print an f32 with ryu: plain notation while the decimal point sits within
        (-6, 13] digits of the first digit, `1.234e33` style otherwise."""
        digits, exp10 = _shortest_f32(self.e_value)
        sign = "-" if digits.startswith("-") else ""
        d = digits.lstrip("-")
        kk = len(d) + exp10
        if 0 <= exp10 and kk <= 13:
            return sign + d + "0" * exp10 + ".0"
        if 0 < kk <= 13:
            return sign + d[:kk] + "." + d[kk:]
        if -6 < kk <= 0:
            return sign + "0." + "0" * (-kk) + d
        e = kk - 1
        return sign + (d if len(d) == 1 else d[0] + "." + d[1:]) + "e" + str(e)

    def as_items(self):
        """(camelCase key, JSON literal) in serde's field order, subject_reads reduced to its file name
        (write_blutils_output.rs:113-124)."""
        return [("isConfig", "true" if self.is_config else "false"), ("runId", json.dumps(str(self.run_id))),
                ("blutilsVersion", json.dumps(self.blutils_version)),
                ("subjectReads", json.dumps(os.path.basename(self.subject_reads.rstrip("/")))),
                ("taxon", json.dumps(self.taxon)), ("outFormat", json.dumps(self.out_format)),
                ("maxTargetSeqs", str(self.max_target_seqs)), ("percIdentity", str(self.perc_identity)),
                ("queryCov", str(self.query_cov)), ("strand", json.dumps(self.strand)), ("eValue", self.e_value_json()),
                ("wordSize", str(self.word_size))]

    def render(self, out_format: str) -> str:
        """The config as it stands in the result document (include/blu_pipeline.h: blu_build_consensus_identities_cfg)."""
        items = self.as_items()
        if out_format == "json":                 # serde_json::to_string_pretty, nested one level deep
            return "{\n" + ",\n".join(f'    "{k}": {v}' for k, v in items) + "\n  }"
        if out_format in ("json-compact", "jsonl"):
            return "{" + ",".join(f'"{k}":{v}' for k, v in items) + "}"
        if out_format == "yaml":                 # serde_yaml block mapping (scalars that need no quotes here)
            return "".join(f"  {k}: {json.loads(v) if v.startswith(chr(34)) else v}\n" for k, v in items)
        raise ValueError(out_format)


@dataclass
class Sequence:
    header: str
    sequence: str

    def blast_header(self) -> str:
        return self.header.split()[0]

    def to_fasta(self) -> str:
        return f">{self.header}\n{self.sequence}\n"


def sequence_content(source: str) -> List[Sequence]:
    """Multi-FASTA from a file or `-` (stdin); multi-line sequences are joined, `>` is stripped from the header
    wherever it occurs (`line.replace(">", "")`), a sequence before any header is an error."""
    text = sys.stdin.read() if source == "-" else open(source).read()
    out: List[Sequence] = []
    header, seq = "", ""
    for line in text.split("\n"):
        if line.endswith("\r"):
            line = line[:-1]                     # BufRead::lines strips "\r\n"
        if not line:
            continue
        if line.startswith(">"):
            if header:
                out.append(Sequence(header, seq))
                seq = ""
            elif seq:
                raise BlastError("unexpected sequence without header")
            header = line.replace(">", "")
        else:
            seq += line
    if header and seq:
        out.append(Sequence(header, seq))
    return out


def validate_blast_database(path: str) -> None:
    """A `<stem>*.nsq` next to the database prefix must exist."""
    stem = os.path.splitext(os.path.basename(path))[0]
    parent = os.path.expanduser(os.path.dirname(path))
    if not sorted(glob.glob(os.path.join(parent, stem + "*.nsq"))):
        raise BlastError(f'Blast database not found: "{path}"')


class ExecuteBlastnProcRepository:
    """adapters/proc/src/execute_blast.rs: `blastn` with the query on stdin; executable overridable for tests."""

    def __init__(self, executable: str = "blastn"):
        self.executable = executable

    def run(self, query_sequences: str, blast_config: BlastBuilder, threads: int):
        cmd = [self.executable, "-db", blast_config.subject_reads, "-outfmt", blast_config.out_format,
               "-max_target_seqs", str(blast_config.max_target_seqs), "-perc_identity", str(blast_config.perc_identity),
               "-qcov_hsp_perc", str(blast_config.query_cov), "-strand", blast_config.strand,
               "-evalue", blast_config.e_value_text(), "-word_size", str(blast_config.word_size),
               "-num_threads", str(threads)]
        try:
            p = subprocess.run(cmd, input=query_sequences, capture_output=True, text=True)
        except OSError as e:
            raise BlastError(f"Unexpected error detected on execute blast: {e}") from None
        return (p.returncode == 0), (p.stdout if p.returncode == 0 else p.stderr)


def run_parallel_blast(input_sequences: str, blast_out_file: str, blast_config: BlastBuilder, blast_execution_repo,
                       overwrite: bool, threads: int):
    """-> (output_file, headers).  Chunks of 50 sequences, `threads` BLAST processes at a time; the chunk outputs are
    appended in chunk order (the reference appends them as they finish: any order of whole chunks)."""
    validate_blast_database(blast_config.subject_reads)
    out_path = os.path.splitext(blast_out_file)[0] + ".out"
    out_dir = os.path.dirname(out_path)
    if out_dir and not os.path.exists(out_dir):
        os.mkdir(out_dir)
    if os.path.exists(out_path):
        if not overwrite:
            raise SystemExit(f'Could not overwrite existing file "{out_path}" when overwrite option is `false`.')
        os.remove(out_path)
    seqs = sequence_content(input_sequences)
    headers = [s.blast_header() for s in seqs]
    chunks = [seqs[i:i + CHUNK_SIZE] for i in range(0, len(seqs), CHUNK_SIZE)]

    def one(chunk):
        return blast_execution_repo.run("".join(s.to_fasta() for s in chunk), blast_config, threads)

    with ThreadPoolExecutor(max_workers=max(1, int(threads))) as pool, open(out_path, "a") as f:
        for index, (ok, text) in enumerate(pool.map(one, chunks)):
            if not ok:
                raise BlastError(f"Unexpected error on process chunk {index}: {text}")
            f.write(text)
    return out_path, headers


def run_blast_and_build_consensus(input_sequences: str, input_taxonomies: str, blast_out_file: str,
                                  blutils_out_file: Optional[str], blast_config: BlastBuilder, blast_execution_repo,
                                  overwrite: bool, threads: int, strategy: str, use_taxid: Optional[bool],
                                  out_format: str = "json", custom_taxon_values: Optional[dict] = None, device: int = 0,
                                  lenient: bool = False):
    """-> the document text (also written to blutils_out_file with the format's extension, or to stdout)."""
    output_file, headers = run_parallel_blast(input_sequences, blast_out_file, blast_config, blast_execution_repo,
                                              overwrite, threads)
    to_file = blutils_out_file is not None
    fmt = out_format if (to_file or out_format != "json") else "json-compact"
    text, _ = pipeline.build_consensus_identities(output_file, input_taxonomies, blast_config.taxon, strategy, use_taxid,
                                                  custom_taxon_values, headers=headers, out_format=fmt, device=device,
                                                  lenient=lenient, parse=False, config=blast_config)
    if to_file:
        path = os.path.splitext(blutils_out_file)[0] + "." + out_format
        if os.path.exists(path):
            os.remove(path)
        parent = os.path.dirname(path)
        if parent and not os.path.exists(parent):
            os.makedirs(parent)
        with open(path, "w") as f:
            f.write(text)
    else:
        sys.stdout.write(text)
    return text
