"""Command-line surface of run_apples.py (apples/OptionsBasic.py:15-70, apples/OptionsRun.py:10-110):
same flags, defaults and cross-flag validation, plus the device options of this build."""
import logging
from multiprocessing import cpu_count
from optparse import OptionParser

from . import __version__

APPLES_COMPAT_VERSION = '2.0.11'


def build_parser():
    p = OptionParser()
    # OptionsBasic
    p.add_option('-t', '--tree', dest='tree_fp', help='path to the reference tree', metavar='FILE')
    p.add_option('-o', '--output', dest='output_fp', help='path for the output jplace file', metavar='FILE')
    p.add_option('-s', '--ref', dest='ref_fp', metavar='FILE',
                 help='path to the reference alignment file (FASTA), containing reference sequences')
    p.add_option('-p', '--protein', dest='protein_seqs', action='store_true', default=False,
                 help='input sequences are protein sequences')
    p.add_option('-T', '--threads', dest='num_thread', type=int, default=0, metavar='NUMBER',
                 help='accepted for compatibility (host-side parsing only); 0 = all cores')
    p.add_option('-f', '--filter', dest='filt_threshold', type=float, default=0.2, metavar='NUMBER',
                 help='ignores distances higher than the given threshold')
    p.add_option('-D', '--disable-reestimation', dest='disable_reestimation', action='store_true', default=False,
                 help='disables branch length reestimation of the backbone tree')
    p.add_option('--debug', dest='debug_mode', action='store_true', default=False, help='Enables debug mode.')
    p.add_option('-v', '--version', dest='print_version', action='store_true', default=False,
                 help='print version number')
    # OptionsRun
    p.add_option('-a', '--database', dest='database_fp', metavar='FILE', help='path to the APPLES database')
    p.add_option('-d', '--distances', dest='dist_fp', metavar='FILE', help='path to the table of observed distances')
    p.add_option('-x', '--extendedref', dest='extended_ref_fp', metavar='FILE',
                 help='path to the extended reference alignment file (FASTA), containing reference and query sequences')
    p.add_option('-q', '--query', dest='query_fp', metavar='FILE',
                 help='path to the query alignment file (FASTA), containing query sequences')
    p.add_option('-m', '--method', dest='method_name', default='FM', metavar='METHOD',
                 help='name of the weighted least squares method (OLS, FM, BME, or BE)')
    p.add_option('-c', '--criterion', dest='criterion_name', default='MLSE', metavar='CRITERIA',
                 help='name of the placement selection criterion (MLSE, ME, or HYBRID)')
    p.add_option('-n', '--negative', dest='negative_branch', action='store_true',
                 help='relaxes positivity constraint on new branch lengths')
    p.add_option('-b', '--base', dest='base_observation_threshold', type=int, default=25, metavar='NUMBER',
                 help='minimum number of observations kept for each query ignoring the filter threshold')
    p.add_option('-V', '--overlap', dest='minimum_alignment_overlap', type=float, default=0.001, metavar='NUMBER',
                 help='minimum fraction of nongap sites needed for a valid pairwise distance')
    p.add_option('-X', '--mask', dest='mask_lowconfidence', action='store_true', default=False,
                 help='masks low confidence (lowercase) characters in the alignments')
    p.add_option('--exclude', dest='exclude_intplace', action='store_true', default=False,
                 help='exclude queries placed on the internal nodes in jplace file')
    # this build
    p.add_option('--clusters', dest='clusters_fp', metavar='FILE',
                 help='TreeCluster output (name<TAB>cluster) to use for the reduced reference instead of the '
                      'built-in max-diameter clustering at 1.2 x the filter threshold')
    p.add_option('--no-clusters', dest='no_clusters', action='store_true', default=False,
                 help='every reference sequence is its own cluster (no reduced reference)')
    p.add_option('--fasttree', dest='fasttree_fp', metavar='FILE',
                 help='FastTree executable for the backbone branch length reestimation (the reference bundles one; '
                      'this build looks for $APPLES_FASTTREE, then FastTree on PATH, and without one -- or with '
                      '"native" -- estimates the same minimum-evolution lengths on the GPU)')
    p.add_option('--gpus', dest='num_gpus', type=int, default=1, metavar='NUMBER',
                 help='number of MI355X devices to shard the queries over (0 = all visible)')
    return p


def options_config(argv=None):
    """Parse + validate as apples/OptionsBasic.py:72-92 and apples/OptionsRun.py:86-110 (same errors, same
    warnings, same precedence of -t over the database's tree and of -d over its sequences)."""
    parser = build_parser()
    options, args = parser.parse_args(argv)
    if options.print_version:
        print('APPLES version %s (apples-mi355x %s)' % (APPLES_COMPAT_VERSION, __version__), flush=True)
        raise SystemExit(0)
    options.reestimate_backbone = not options.disable_reestimation
    if options.debug_mode:
        logging.getLogger().setLevel(logging.DEBUG)
    if not options.num_thread:
        options.num_thread = cpu_count()
    if options.dist_fp:
        options.reestimate_backbone = False
        if options.ref_fp:
            raise ValueError('Input should be either an alignment or a distance matrix, but not both!')
        if options.database_fp:  # apples/OptionsRun.py:92-97
            logging.warning('Input contains both an APPLES database and a distance matrix. Database sequences '
                            'will be ignored. Database phylogeny will be used if user did not provide a phylogeny '
                            '(using -t option). ')
    if options.database_fp:  # apples/OptionsRun.py:99-106
        if options.ref_fp:
            raise ValueError('Input should be either an alignment or a APPLES database file, but not both!')
        if options.tree_fp:
            logging.warning('Input contains both an APPLES database and a tree file. User provided tree has '
                            'higher priority and therefore will be used.')
    if not options.tree_fp and not options.database_fp:
        raise ValueError('No input backbone tree provided by user.')
    if options.query_fp and options.extended_ref_fp:
        raise ValueError('Input should be either an extended alignment or a query alignment, but not both!')
    return options, args
