"""Command line with the reference's flags, short flags and defaults (reference mustache/main.py:66-117 for the two hot-path
commands; the CLI defaults are authoritative, not the class defaults or the help texts).

    python -m mustache_amd.main findflanks BAM -o out.findflanks.tsv
    python -m mustache_amd.main pairflanks FLANKS BAM GENOME -o out.pairflanks.tsv

The Snakemake workflow of the reference calls exactly these two commands (workflow/Snakefile:82-87, 105-109); every flag
it passes is accepted here.  Beside them the SURVEY.md section 8(f) "next" rows are registered: ``recall`` (main.py:272-284),
``inferseq-overlap`` (167-182), ``inferseq-reference`` (143-164), ``inferseq-database`` (201-214), ``formatbam`` (262-269).  The rest of the toolbox is outside the hot path.
"""
from __future__ import annotations

import sys
from os.path import abspath

import click


def log_params(**kw):
    print("[mustache_amd] parameters: " + ", ".join("%s=%s" % (k, v) for k, v in kw.items()), file=sys.stderr)


@click.group()
def cli():
    """MI355X-native flank discovery (drop-in for `mustache findflanks` / `mustache pairflanks`)."""


@cli.command(short_help="Find insertion sites and reconstruct flanks of inserted sequence")
@click.argument("bamfile", type=click.Path(exists=True))
@click.option("--min_softclip_length", "-minlen", default=8)
@click.option("--min_softclip_count", "-mincount", default=2)
@click.option("--min_alignment_quality", "-minq", default=20)
@click.option("--min_alignment_inner_length", "-minial", default=21)
@click.option("--min_distance_to_mate", "-mindist", default=22)
@click.option("--min_softclip_ratio", "-minratio", default=0.10)
@click.option("--max_indel_ratio", "-maxir", default=0.03)
@click.option("--large_insertion_cutoff", "-lins", default=30)
@click.option("--min_count_consensus", "-mcc", default=2)
@click.option("--sample_id", "-id", default=None)
@click.option("--output_file", "-o", default="mustache.findflanks.tsv")
@click.option("--exclude-flags", "exclude_flags", default="0", help="Skip records with any of these FLAG bits, e.g. 0x900 for a BAM that did not go through formatbam (not in the reference).")
@click.option("--decoder", type=click.Choice(["gpu", "native", "python"]), default="gpu", help="Who reads the BAM: the GPU (inflate + record parse + packing as kernels), host threads (libmfkio.so) or the Python statement of the packing rules; same arrays (not in the reference).")
@click.option("--device", default="cuda:0", help="GPU to run on (not in the reference).")
def findflanks(bamfile, min_softclip_length, min_softclip_count, min_alignment_quality, min_alignment_inner_length,
               min_distance_to_mate, min_softclip_ratio, max_indel_ratio, large_insertion_cutoff,
               min_count_consensus, sample_id, output_file, exclude_flags, decoder, device):
    from .findflanks import _findflanks
    if sample_id is None:
        sample_id = abspath(output_file)           # reference main.py:84-85
    log_params(command="findflanks", bamfile=bamfile, min_softclip_length=min_softclip_length,
               min_softclip_count=min_softclip_count, min_alignment_quality=min_alignment_quality,
               min_alignment_inner_length=min_alignment_inner_length, min_distance_to_mate=min_distance_to_mate,
               min_softclip_ratio=min_softclip_ratio, max_indel_ratio=max_indel_ratio,
               large_insertion_cutoff=large_insertion_cutoff, min_count_consensus=min_count_consensus,
               sample_id=sample_id, output_file=output_file)
    _findflanks(bamfile, min_softclip_length, min_softclip_count, min_alignment_quality, min_alignment_inner_length,
                min_distance_to_mate, min_softclip_ratio, max_indel_ratio, large_insertion_cutoff,
                min_count_consensus, sample_id, output_file, device=device, exclude_flags=int(str(exclude_flags), 0), decoder=decoder)


@cli.command(short_help="Pair identified flanks with each other to represent 5' and 3' ends of inserted sequence.")
@click.argument("flanksfile", type=click.Path(exists=True))
@click.argument("bamfile", type=click.Path(exists=True))
@click.argument("genome", type=click.Path(exists=True))
@click.option("--max_direct_repeat_length", "-maxdr", default=20)
@click.option("--min_alignment_quality", "-minq", default=20)
@click.option("--min_alignment_inner_length", "-minial", default=21)
@click.option("--max_junction_spanning_prop", "-maxjsp", default=0.15)
@click.option("--large_insertion_cutoff", "-lins", default=30)
@click.option("--output_file", "-o", default="mustache.pairflanks.tsv")
@click.option("--exclude-flags", "exclude_flags", default="0", help="Skip records with any of these FLAG bits, e.g. 0x900 for a BAM that did not go through formatbam (not in the reference).")
@click.option("--device", default="cuda:0", help="GPU to run on (not in the reference).")
@click.option("--ir-backend", "ir_backend", type=click.Choice(["builtin", "einverted"]), default="builtin",
              help="Inverted-repeat detector: the library's own (default) or EMBOSS einverted as the reference runs it "
                   "(needs the executable on PATH; not in the reference).  PARITY NOTE: with 'builtin' the columns has_IR, IR_length, "
                   "IR_5p and IR_3p follow this build's restatement of einverted's model (DESIGN.md section 6), which could not be checked "
                   "against EMBOSS (absent from the build image and the reference tree); every other column is pinned to the reference. "
                   "Use 'einverted' for reference-identical IR columns.")
def pairflanks(flanksfile, bamfile, genome, max_direct_repeat_length, min_alignment_quality,
               min_alignment_inner_length, max_junction_spanning_prop, large_insertion_cutoff, output_file, exclude_flags, device,
               ir_backend):
    from .pairflanks import _pairflanks
    log_params(command="pairflanks", flanksfile=flanksfile, bamfile=bamfile, genome=genome,
               max_direct_repeat_length=max_direct_repeat_length, min_alignment_quality=min_alignment_quality,
               min_alignment_inner_length=min_alignment_inner_length,
               max_junction_spanning_prop=max_junction_spanning_prop, large_insertion_cutoff=large_insertion_cutoff,
               output_file=output_file)
    _pairflanks(flanksfile, bamfile, genome, max_direct_repeat_length, min_alignment_quality,
                min_alignment_inner_length, max_junction_spanning_prop, large_insertion_cutoff, output_file,
                device=device, ir_backend=ir_backend, exclude_flags=int(str(exclude_flags), 0))


@cli.command(short_help="Formats a BAM file for use with mustache. Usually not necessary, unless using the experimental extendpairs command.")
@click.argument("in_sam", type=click.Path(exists=True))
@click.argument("out_bam")
@click.option("--single-end", is_flag=True, default=False, help="Add this flag for single-end files. default=False")
@click.option("--keep-tmp-files", is_flag=True, default=False, help="Add this flag if you want to keep intermediate temporary files. default=False")
def formatbam(in_sam, out_bam, single_end, keep_tmp_files):
    """Reference main.py:262-269: secondary / supplementary alignments out, mate tags, coordinate sort, index."""
    from .formatbam import _formatbam
    log_params(command="formatbam", in_sam=in_sam, out_bam=out_bam, single_end=single_end, keep_tmp_files=keep_tmp_files)
    _formatbam(in_sam, out_bam, single_end, keep_tmp_files)


@cli.command(short_help="Recall softclip counts and runthrough counts from BAM file at specified pairflank insertions.")
@click.argument("pairsfile", type=click.Path(exists=True))
@click.argument("bamfile", type=click.Path(exists=True))
@click.option("--min_alignment_quality", "-minq", default=20)
@click.option("--min_alignment_inner_length", "-minial", default=21)
@click.option("--large_insertion_cutoff", "-lins", default=30)
@click.option("--output_file", "-o", default="mustache.recall.tsv")
@click.option("--device", default="cuda:0", help="GPU to run on (not in the reference).")
def recall(pairsfile, bamfile, min_alignment_quality, min_alignment_inner_length, large_insertion_cutoff, output_file, device):
    """reference main.py:272-284"""
    from .recall import _recall
    log_params(command="recall", pairsfile=pairsfile, bamfile=bamfile, min_alignment_quality=min_alignment_quality,
               min_alignment_inner_length=min_alignment_inner_length, large_insertion_cutoff=large_insertion_cutoff,
               output_file=output_file)
    _recall(pairsfile, bamfile, min_alignment_quality, min_alignment_inner_length, large_insertion_cutoff, output_file,
            device=device)


@cli.command(name="inferseq-overlap", short_help="Infers the identity of an inserted sequence by checking if the two flanks overlap with one another.")
@click.argument("pairsfile", type=click.Path(exists=True))
@click.option("--min_overlap_score", "-minscore", default=10)
@click.option("--min_overlap_perc_identity", "-minopi", default=0.9)
@click.option("--min_inferseq_size", "-minsize", default=30)
@click.option("--output_file", "-o", default="mustache.inferseq_overlap.tsv")
@click.option("--device", default="cuda:0", help="GPU to run on (not in the reference).")
def inferseq_overlap(pairsfile, min_overlap_score, min_overlap_perc_identity, min_inferseq_size, output_file, device):
    """reference main.py:167-182"""
    from .inferseqoverlap import _inferseq_overlap
    log_params(command="inferseq_overlap", pairsfile=pairsfile, min_overlap_score=min_overlap_score,
               min_overlap_perc_identity=min_overlap_perc_identity, min_inferseq_size=min_inferseq_size,
               output_file=output_file)
    _inferseq_overlap(pairsfile, min_overlap_score, min_overlap_perc_identity, min_inferseq_size, output_file, device=device)


@cli.command(name="inferseq-reference", short_help="Infers the identity of an inserted sequence by aligning flank pairs to a reference genome.")
@click.argument("pairsfile", type=click.Path(exists=True))
@click.argument("inferseq_reference", type=click.Path(exists=True))
@click.option("--min_perc_identity", "-minident", default=0.95)
@click.option("--max_internal_softclip_prop", "-maxclip", default=0.05)
@click.option("--max_inferseq_size", "-maxsize", default=200000)
@click.option("--min_inferseq_size", "-minsize", default=30)
@click.option("--keep-intermediate/--no-keep-intermediate", default=False)
@click.option("--output_file", "-o", default="mustache.inferseq_reference.tsv")
@click.option("--device", default="cuda:0", help="GPU to run on (not in the reference).")
@click.option("--alignments-bam", "alignments_bam", type=click.Path(exists=True), default=None,
              help="Flank alignments made beforehand (bowtie2 --local --all, NM and AS tags) instead of running bowtie2 "
                   "here (not in the reference).")
def inferseq_reference(pairsfile, inferseq_reference, min_perc_identity, max_internal_softclip_prop, max_inferseq_size,
                       min_inferseq_size, keep_intermediate, output_file, device, alignments_bam):
    """reference main.py:143-164"""
    from .inferseq import _inferseq_reference
    log_params(command="inferseq_reference", pairsfile=pairsfile, inferseq_reference=inferseq_reference,
               min_perc_identity=min_perc_identity, max_internal_softclip_prop=max_internal_softclip_prop,
               max_inferseq_size=max_inferseq_size, min_inferseq_size=min_inferseq_size,
               keep_intermediate=keep_intermediate, output_file=output_file)
    _inferseq_reference(pairsfile, inferseq_reference, min_perc_identity, max_internal_softclip_prop, max_inferseq_size,
                        min_inferseq_size, keep_intermediate, output_file, device=device, alignments_bam=alignments_bam)


@cli.command(name="inferseq-database", short_help="Infers the identity of an inserted sequence by aligning flank pairs to an database of known inserted elements.")
@click.argument("pairsfile", type=click.Path(exists=True))
@click.argument("inferseq_database", type=click.Path(exists=True))
@click.option("--min_perc_identity", "-minident", default=0.90)
@click.option("--max_internal_softclip_prop", "-maxclip", default=0.05)
@click.option("--max_edge_distance", "-maxedgedist", default=10)
@click.option("--output_file", "-o", default="mustache.inferseq_database.tsv")
@click.option("--keep-intermediate/--no-keep-intermediate", default=False)
@click.option("--device", default="cuda:0", help="GPU to run on (not in the reference).")
@click.option("--alignments-bam", "alignments_bam", type=click.Path(exists=True), default=None,
              help="Flank alignments made beforehand (BAM with NM and AS tags) instead of running bowtie2 here (not in the reference).")
def inferseq_database(pairsfile, inferseq_database, min_perc_identity, max_internal_softclip_prop, max_edge_distance, output_file,
                      keep_intermediate, device, alignments_bam):
    """Reference main.py:201-214."""
    from .inferseqdatabase import _inferseq_database
    log_params(command="inferseq_database", pairsfile=pairsfile, inferseq_database=inferseq_database, min_perc_identity=min_perc_identity,
               max_internal_softclip_prop=max_internal_softclip_prop, max_edge_distance=max_edge_distance, output_file=output_file,
               keep_intermediate=keep_intermediate)
    _inferseq_database(pairsfile, inferseq_database, min_perc_identity, max_internal_softclip_prop, max_edge_distance, output_file,
                       keep_intermediate, device=device, alignments_bam=alignments_bam)


if __name__ == "__main__":
    cli()
