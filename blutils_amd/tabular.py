"""`blu blastn build-tabular`: a blutils result document -> the 12-column table.

Mirror of the reference use-case (same name, arguments and quirks):

    parse_consensus_as_tabular(blutils_result, output_file, result_format)
    core/src/use_cases/parse_consensus_as_tabular/mod.rs:15-173

Host-only text work (no GPU, no native code): it is the step after the result writer, kept byte-compatible so
its output can be diffed against the reference's.  Reference behaviour kept on purpose:

* the existence check looks for the argument with its extension replaced by `.json`, whatever the input format
  (mod.rs:24-32);
* with an output file the rows are appended WITHOUT a line terminator (`write_or_append_to_file` writes the bytes
  it is given, shared/write_or_append_to_file.rs:23-37; only `println!` on the stdout path adds one), except the
  `query<TAB>null` row of a query without consensus, which carries its own `\\n` (mod.rs:112) — so on stdout that
  row is followed by an empty line;
* numbers are printed like Rust's `Display for f64` (shortest round-trip digits, never an exponent, no `.0`).
"""
from __future__ import annotations

import json
import os
import sys
import uuid
from decimal import Decimal
from typing import Optional

HEADER = ("run-id", "query", "type", "rank", "identifier", "perc-identity", "bit-score", "taxonomy", "mutated",
          "single-match", "occurrences", "accessions")


class TabularError(Exception):
    """use_case_err(...) of the reference."""


def rust_f64(x) -> str:
    """`format!("{}", x)` for an f64."""
    x = float(x)
    if x != x:
        return "NaN"
    if x in (float("inf"), float("-inf")):
        return "inf" if x > 0 else "-inf"
    r = repr(x)
    if "e" in r or "E" in r:
        r = format(Decimal(r), "f")
    if r.endswith(".0"):
        r = r[:-2]
    return r


def full_rank_string(rank) -> str:
    """LinnaeanRank::as_full_rank_string (linnaean_ranks.rs:92-106) of a deserialized `reachedRank` / bean `rank`.
    The enum is camelCase with an untagged `Other(String)` (linnaean_ranks.rs:14-29): on the wire every rank is a
    plain string, a variant name reads back as that variant and prints as the same word, anything else is `Other`
    and prints as is."""
    if not isinstance(rank, str):
        raise TabularError(f"unable to parse content: invalid rank `{rank}`")
    return rank


def _yaml_load(text: str):
    import yaml
    return yaml.safe_load(text)


def _read(source: str) -> str:
    if source == "-":
        return sys.stdin.read()
    with open(source) as f:
        return f.read()


def load_content(blutils_result: str, result_format: str) -> dict:
    """FileOrStdin::{json_content, json_line_content, yaml_content} (file_or_stdin.rs:101-170) -> BlutilsOutput."""
    text = _read(blutils_result)
    try:
        if result_format == "json":
            doc = json.loads(text)
        elif result_format == "yaml":
            doc = _yaml_load(text)
        elif result_format == "jsonl":
            doc = {"results": [], "config": None}
            for line in text.split("\n"):
                if not line:
                    continue
                if "isConfig" in line:
                    doc["config"] = json.loads(line)
                else:
                    doc["results"].append(json.loads(line))
        else:
            raise TabularError(f"unknown format `{result_format}`")
    except (ValueError, ImportError) as e:
        kind = {"json": "content as JSON", "yaml": "content as YAML", "jsonl": "line as JSON"}[result_format]
        raise TabularError(f"unable to parse {kind}: {e}") from None
    except Exception as e:  # yaml.YAMLError
        if result_format == "yaml":
            raise TabularError(f"unable to parse content as YAML: {e}") from None
        raise
    if not isinstance(doc, dict) or not isinstance(doc.get("results"), list):
        raise TabularError("unable to parse content: missing field `results`")
    for r in doc["results"]:
        # serde refuses anything that is not a QueryWithConsensus — e.g. the `null` config line that
        # write_blutils_output puts first in a JSONL file when there is no config (write_blutils_output.rs:166-176):
        # the reference cannot read that file back, and neither does this
        if not isinstance(r, dict) or "query" not in r:
            what = "line as JSON" if result_format == "jsonl" else f"content as {result_format.upper()}"
            raise TabularError(f"unable to parse {what}: invalid type, expected struct QueryWithConsensus")
    return doc


def parse_consensus_as_tabular(blutils_result: str = "-", output_file: Optional[str] = None, result_format: str = "json",
                               stdout=None) -> None:
    if blutils_result != "-":
        probe = os.path.splitext(blutils_result)[0] + ".json"       # PathBuf::set_extension("json")
        if not os.path.exists(probe):
            raise TabularError(f"The file `{blutils_result}` does not exist.")
    content = load_content(blutils_result, result_format)
    to_stdout = output_file is None
    out = stdout if stdout is not None else sys.stdout
    fh = None
    if not to_stdout:
        path = os.path.splitext(output_file)[0] + ".tsv"
        if os.path.exists(path):
            os.remove(path)
        fh = open(path, "a")

    def emit(row: str) -> None:
        if to_stdout:
            out.write(row + "\n")            # println!
        else:
            fh.write(row)                    # the bytes as given

    try:
        emit("\t".join(HEADER))
        cfg = content.get("config")
        run_id = cfg["runId"] if isinstance(cfg, dict) and cfg.get("runId") else str(uuid.uuid4())
        null = "null"
        for result in content["results"]:
            bean = result.get("taxon")
            query = result["query"]
            if bean is None:
                emit(f"{query}\tnull\n")
                continue
            rid = result.get("runId") or run_id
            tax = bean.get("taxonomy")
            emit("\t".join((str(rid), query, "consensus", full_rank_string(bean["reachedRank"]), str(bean["identifier"]),
                            rust_f64(bean["percIdentity"]), rust_f64(bean["bitScore"]), null if tax is None else str(tax),
                            "true" if bean["mutated"] else "false", "true" if bean["singleMatch"] else "false", null, null)))
            for c in bean.get("consensusBeans") or []:
                ctax = c.get("taxonomy")
                emit("\t".join((str(rid), query, "blast-match", full_rank_string(c["rank"]), str(c["identifier"]), null,
                                rust_f64(bean["bitScore"]), null if ctax is None else str(ctax), null, null,
                                str(int(c["occurrences"])), ", ".join(c["accessions"]))))
    finally:
        if fh is not None:
            fh.close()
