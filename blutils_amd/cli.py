"""`blu`-compatible driver for the consensus step.

    python -m blutils_amd.cli blastn build-consensus BLAST_OUT -t TAX.json --taxon bacteria --strategy relaxed
        [-c CUTOFFS.yaml] [-u] [--blutils-out-file OUT] [--out-format json|jsonl|yaml]

Same arguments as the reference's `blu blastn build-consensus`
(ports/cli/src/cmds/blast/commands.rs:105-143, cmds/blast/mod.rs:104-146): without --blutils-out-file the
document goes to stdout (compact JSON, as serde_json::to_writer prints it); with it, the extension is forced to
the format's (write_blutils_output.rs:39-52) and JSON is pretty-printed.

    python -m blutils_amd.cli blastn build-tabular [BLU_RESULT|-] [-o OUT] [-i json|jsonl|yaml]

= `blu blastn build-tabular` (commands.rs:145-161, parse_consensus_as_tabular/mod.rs:15).

    python -m blutils_amd.cli cache-db TAX.json CACHE [-u]

writes the binary cache of a taxonomies file (not in the reference CLI; pass CACHE as -t afterwards).  
    python -m blutils_amd.cli blastn run-with-consensus [QUERY.fa|-] -d DB -t TAX.json --blast-out-file B --taxon T
        --strategy S [...]

= `blu blastn run-with-consensus` (commands.rs:24-103): `blastn` itself stays an external process (blutils_amd/blast.py).
The DB builders are not part of this engine."""
from __future__ import annotations

import argparse
import os
import sys

from . import blast, pipeline, tabular


def build_parser() -> argparse.ArgumentParser:
    ap = argparse.ArgumentParser(prog="blu", description="MI355X-native consensus step of blutils")
    sub = ap.add_subparsers(dest="cmd", required=True)
    blastn = sub.add_parser("blastn").add_subparsers(dest="sub", required=True)
    bc = blastn.add_parser("build-consensus", help="consensus identities from a BLAST outfmt-6 table")
    bc.add_argument("blast_out")
    bc.add_argument("-t", "--tax-file", required=True)
    bc.add_argument("--blutils-out-file")
    bc.add_argument("--taxon", required=True, choices=["fungi", "bacteria", "eukaryotes", "custom"])
    bc.add_argument("-c", "--custom-taxon-cutoff-file")
    bc.add_argument("--strategy", required=True, choices=["cautious", "relaxed"])
    bc.add_argument("-u", "--use-taxid", action="store_true")
    bc.add_argument("--out-format", default="json", choices=["json", "jsonl", "yaml"])
    bc.add_argument("--device", type=int, default=0, help="HIP device ordinal (not in the reference CLI)")
    rw = blastn.add_parser("run-with-consensus", help="blastn fan-out (chunks of 50 queries) + consensus")
    rw.add_argument("query", nargs="?", default="-")
    rw.add_argument("-d", "--database", required=True)
    rw.add_argument("-t", "--tax-file", required=True)
    rw.add_argument("--blast-out-file", required=True)
    rw.add_argument("--blutils-out-file")
    rw.add_argument("--out-format", default="json", choices=["json", "jsonl", "yaml"])
    rw.add_argument("--taxon", required=True, choices=["fungi", "bacteria", "eukaryotes", "custom"])
    rw.add_argument("-c", "--custom-taxon-cutoff-file")
    rw.add_argument("--strategy", required=True, choices=["cautious", "relaxed"])
    rw.add_argument("-u", "--use-taxid", action="store_true")
    rw.add_argument("-f", "--force-overwrite", action="store_true")
    rw.add_argument("-m", "--max-target-seqs", type=int)
    rw.add_argument("-p", "--perc-identity", type=int)
    rw.add_argument("-q", "--query-cov", type=int)
    rw.add_argument("--strand", choices=["both", "plus", "minus"])
    rw.add_argument("-e", "--e-value", type=float)
    rw.add_argument("-w", "--word-size", type=int)
    rw.add_argument("--threads", type=int, default=1, help="the reference's global `--threads` option (default 1)")
    rw.add_argument("--blastn", default="blastn", help="blastn executable (not in the reference CLI)")
    rw.add_argument("--device", type=int, default=0, help="HIP device ordinal (not in the reference CLI)")
    bt = blastn.add_parser("build-tabular", help="blutils result document -> TSV")
    bt.add_argument("blu_result", nargs="?", default="-")
    bt.add_argument("-o", "--output-file")
    bt.add_argument("-i", "--input-format", default="json", choices=["json", "jsonl", "yaml"])
    cd = sub.add_parser("cache-db", help="binary cache of a *.blutils.json (pass it as --tax-file afterwards)")
    cd.add_argument("tax_file")
    cd.add_argument("cache_file")
    cd.add_argument("-u", "--use-taxid", action="store_true")
    return ap


def _run_with_consensus(args) -> int:
    """ports/cli/src/cmds/blast/mod.rs:24-102"""
    config = blast.BlastBuilder.default(args.database, args.taxon)
    if args.max_target_seqs is not None:
        config = config.with_max_target_seqs(args.max_target_seqs)
    if args.perc_identity is not None:
        config = config.with_perc_identity(args.perc_identity)
    if args.query_cov is not None:
        config = config.with_query_cov(args.query_cov)
    if args.strand is not None:
        config = config.with_strand(args.strand)
    if args.e_value is not None:
        config = config.with_e_value(args.e_value)
    if args.word_size is not None:
        config = config.with_word_size(args.word_size)
    custom = None
    if args.custom_taxon_cutoff_file:
        custom = pipeline.custom_taxon_from_file(args.custom_taxon_cutoff_file)
    elif args.taxon == "custom":
        raise SystemExit("Custom taxon values are required when the custom taxon option is selected.")
    try:
        blast.run_blast_and_build_consensus(args.query, args.tax_file, args.blast_out_file, args.blutils_out_file, config,
                                            blast.ExecuteBlastnProcRepository(args.blastn), args.force_overwrite,
                                            args.threads, args.strategy, args.use_taxid, args.out_format, custom,
                                            device=args.device)
    except blast.BlastError as e:
        raise SystemExit(str(e))
    return 0


def main(argv=None) -> int:
    args = build_parser().parse_args(argv)
    if args.cmd == "cache-db":
        pipeline.build_db_cache(args.tax_file, args.cache_file, args.use_taxid)
        return 0
    if args.sub == "build-tabular":
        try:
            tabular.parse_consensus_as_tabular(args.blu_result, args.output_file, args.input_format)
        except tabular.TabularError as e:
            raise SystemExit(str(e))
        return 0
    if args.sub == "run-with-consensus":
        return _run_with_consensus(args)
    custom = None
    if args.custom_taxon_cutoff_file:
        custom = pipeline.custom_taxon_from_file(args.custom_taxon_cutoff_file)        # CustomTaxon::from_file
    elif args.taxon == "custom":
        # cmds/blast/mod.rs:114-117
        raise SystemExit("Custom taxon values are required when the custom taxon option is selected.")
    to_file = args.blutils_out_file is not None
    fmt = args.out_format if (to_file or args.out_format != "json") else "json-compact"
    if to_file:
        path = os.path.splitext(args.blutils_out_file)[0] + "." + args.out_format      # PathBuf::set_extension
        parent = os.path.dirname(path)
        if parent and not os.path.exists(parent):
            os.makedirs(parent)
        pipeline.build_consensus_identities(args.blast_out, args.tax_file, args.taxon, args.strategy, args.use_taxid, custom,
                                            headers=None, out_format=fmt, device=args.device, parse=False, out_path=path)
    else:
        text, _ = pipeline.build_consensus_identities(args.blast_out, args.tax_file, args.taxon, args.strategy,
                                                      args.use_taxid, custom, headers=None, out_format=fmt,
                                                      device=args.device, parse=False)
        sys.stdout.write(text)
    return 0


if __name__ == "__main__":
    sys.exit(main())
