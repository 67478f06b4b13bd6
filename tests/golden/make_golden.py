#!/usr/bin/env python3
"""Distil the reference's only golden artefact into a small committed fixture.

Source (read-only, build container only):
  /root/reference/test/mock/output/zymo-mock/blutils.consensus.json
    — a real `blu blastn run-with-consensus` output (blutils 7.1.3, taxon
      bacteria, text lineages, maxTargetSeqs 50; 3626 queries, 2283 with a taxon).
  /root/reference/docs/book/02_run_blast_and_generate_consensus_identities.md:192-249
    — the worked example (blutils 8.3.1, bacteria, relaxed).
  /root/reference/assets/custom-taxon-cutoffs-bacteria-16S.yaml

The 2283 results collapse to 253 distinct `taxon` objects; the fixture keeps
each distinct object once with the number of queries that carry it.  The
fixture holds DATA only (expected outputs); the hit tables are re-synthesised
from the consensus beans by tests/golden_recipe.py (SURVEY §8c recipe).

Run:  python tests/golden/make_golden.py      (needs /root/reference)
"""
import gzip
import json
import os
import re

REF = "/root/reference"
HERE = os.path.dirname(os.path.abspath(__file__))


def main():
    src = os.path.join(REF, "test/mock/output/zymo-mock/blutils.consensus.json")
    d = json.load(open(src))
    uniq = {}
    n_null = 0
    for r in d["results"]:
        t = r.get("taxon")
        if not t:
            n_null += 1
            continue
        k = json.dumps(t, sort_keys=True)
        e = uniq.setdefault(k, {"taxon": t, "n_queries": 0, "example_query": r["query"]})
        e["n_queries"] += 1
    out = {
        "source": "test/mock/output/zymo-mock/blutils.consensus.json",
        "config": d["config"],
        "n_results": len(d["results"]),
        "n_without_taxon": n_null,
        "cases": sorted(uniq.values(), key=lambda e: e["example_query"]),
    }
    with gzip.open(os.path.join(HERE, "zymo_mock_distilled.json.gz"), "wt", compresslevel=9) as f:
        json.dump(out, f, sort_keys=True, separators=(",", ":"))

    # The same cases as BYTES: the text of each case's `taxon` object exactly as the reference wrote it
    # (write_blutils_output.rs:138, serde_json::to_string_pretty), for pinning the product writer's layout.
    raw = open(src).read()
    texts = {}
    for m in re.finditer(r'\n      "query": "([^"]*)",\n      "taxon": \{\n', raw):
        start = m.end() - 2                       # the opening brace
        depth, i = 0, start
        while True:
            ch = raw[i]
            if ch == '"':                         # skip a string (no escapes to worry about beyond \" in this file)
                i += 1
                while raw[i] != '"':
                    i += 2 if raw[i] == "\\" else 1
            elif ch == "{":
                depth += 1
            elif ch == "}":
                depth -= 1
                if depth == 0:
                    break
            i += 1
        texts[m.group(1)] = raw[start:i + 1]
    with gzip.open(os.path.join(HERE, "zymo_mock_taxon_text.json.gz"), "wt", compresslevel=9) as f:
        json.dump({"source": out["source"], "indent": "the objects sit at 6 spaces of indentation, as in the document",
                   "text_of": {e["example_query"]: texts[e["example_query"]] for e in out["cases"]}}, f, sort_keys=True)
    for e in out["cases"]:
        assert json.loads(texts[e["example_query"]]) == e["taxon"]

    # worked example from the user guide: the JSON block inside the markdown
    md = open(os.path.join(REF, "docs/book/02_run_blast_and_generate_consensus_identities.md")).read()
    m = re.search(r"```bash\n(\{\n  \"results\": \[.*?\n\})\n```", md, re.S)
    doc = json.loads(m.group(1))
    with open(os.path.join(HERE, "docs_worked_example.json"), "w") as f:
        json.dump({"source": "docs/book/02_run_blast_and_generate_consensus_identities.md:192-249",
                   "strategy": "relaxed", "taxon": doc["config"]["taxon"],
                   "blutilsVersion": doc["config"]["blutilsVersion"],
                   "results": [{"query": r["query"], "taxon": r["taxon"]} for r in doc["results"]]},
                  f, indent=1, sort_keys=True)

    # custom cutoffs asset (8 integers)
    import yaml
    cut = yaml.safe_load(open(os.path.join(REF, "assets/custom-taxon-cutoffs-bacteria-16S.yaml")))
    with open(os.path.join(HERE, "custom_taxon_cutoffs_bacteria_16S.json"), "w") as f:
        json.dump({"source": "assets/custom-taxon-cutoffs-bacteria-16S.yaml", "values": cut}, f, indent=1)
    print(f"{len(uniq)} distinct taxon objects from {len(d['results']) - n_null} results")


if __name__ == "__main__":
    main()
