#!/usr/bin/env python3
"""Builds the APPLES database cache that ``run_apples.py -a`` reads (the reference's
build_applesdtb.py, flags of apples/OptionsBasic.py:15-70): backbone tree + reference alignment ->
indexed tree, extended Newick, reduced reference (clusters at 1.2 x -f, consensus representatives).
Branch lengths are re-estimated first unless -D (apples_amd/reestimate.py: FastTree if installed, else the GPU estimator)."""
import logging
import sys
import time
from optparse import OptionParser

from apples_amd import database, treecluster
from apples_amd.fasta import read_alignment
from apples_amd.reference import ReducedReference, read_treecluster
from apples_amd.tree import extended_newick, read_tree


def main(argv=None):
    start = time.time()
    p = OptionParser()
    p.add_option('-t', '--tree', dest='tree_fp', metavar='FILE', help='path to the reference tree')
    p.add_option('-o', '--output', dest='output_fp', metavar='FILE', help='path for the output APPLES database')
    p.add_option('-s', '--ref', dest='ref_fp', metavar='FILE', help='path to the reference alignment file (FASTA)')
    p.add_option('-p', '--protein', dest='protein_seqs', action='store_true', default=False,
                 help='input sequences are protein sequences')
    p.add_option('-T', '--threads', dest='num_thread', type=int, default=0, metavar='NUMBER', help='accepted for compatibility')
    p.add_option('-f', '--filter', dest='filt_threshold', type=float, default=0.2, metavar='NUMBER',
                 help='ignores distances higher than the given threshold (clusters are cut at 1.2 x this)')
    p.add_option('-D', '--disable-reestimation', dest='disable_reestimation', action='store_true', default=False,
                 help='disables branch length reestimation of the backbone tree')
    p.add_option('--fasttree', dest='fasttree_fp', metavar='FILE', help='FastTree executable for the reestimation')
    p.add_option('--clusters', dest='clusters_fp', metavar='FILE', help='TreeCluster output to use instead of the built-in clustering')
    p.add_option('--no-clusters', dest='no_clusters', action='store_true', default=False,
                 help='every reference sequence is its own cluster')
    options, _ = p.parse_args(argv)
    if not options.tree_fp:
        raise ValueError('No input backbone tree provided by user.')
    if not options.ref_fp:
        raise ValueError('No reference alignment provided by user.')
    if not options.output_fp:
        raise ValueError('No output path provided by user.')
    if not options.disable_reestimation:  # build_applesdtb.py -> prepareTree (apples/prepareTree.py:20-21)
        from apples_amd.reestimate import cleanup, reestimate_backbone
        reestimate_backbone(options)
    tree = read_tree(options.tree_fp)
    if not options.disable_reestimation:
        cleanup(options)
    newick = extended_newick(tree)
    ref = read_alignment(options.ref_fp, options.protein_seqs, False)
    if options.clusters_fp:
        clusters = read_treecluster(options.clusters_fp)
    elif options.no_clusters:
        clusters = None
        if any(n not in tree.name_to_node for n in ref.names):  # only backbone leaves are ever observed
            clusters = [('-1', [n for n in ref.names if n in tree.name_to_node])]
    else:
        clusters = treecluster.grouped(tree, options.filt_threshold * 1.2)
    reference = ReducedReference(ref, options.protein_seqs, clusters)
    database.save(options.output_fp, tree, newick, reference, options.filt_threshold)
    logging.warning('[%s] APPLES database is built in %.3f seconds.' % (time.strftime('%H:%M:%S'), time.time() - start))


if __name__ == '__main__':
    main()
