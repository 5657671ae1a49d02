#!/usr/bin/env python3
"""Drop-in for the reference's run_apples.py on its per-query hot path: same flags, same jplace,
placements computed on MI355X.  -a reads this build's own database cache
(build_applesdtb.py; the reference's pickles of third-party classes cannot be read).  Backbone branch
re-estimation (off with -D) runs an external FastTree when one is found, else this build's GPU estimator of
the same minimum-evolution lengths (apples_amd/reestimate.py)."""
import logging
import sys
import time

import numpy as np

from apples_amd.dismat import read_dismat as read_distance_table, read_dismat_binary, read_dismat_py  # noqa: F401
from apples_amd.fasta import read_alignment
from apples_amd.jplace import iter_text, keep_mask, write_native
from apples_amd.options import options_config
from apples_amd.reference import ReducedReference, read_treecluster
from apples_amd.tree import extended_newick, read_tree
from apples_amd.worker import QueryWorker


def read_dismat(f):
    """(kept under the reference's name) text table from an open file: apples_amd.dismat.read_dismat_py"""
    return read_dismat_py(f)


def main(argv=None):
    startb = time.time()
    options, _ = options_config(argv)
    start = time.time()
    reference = None
    # The input files are read while the tree is parsed (the scanners are native and run without the interpreter lock): at
    # 200 000 x 1 000 the reference alignment takes 0.14 s, the queries 0.06 s, the tree 0.27 s.  Not with backbone
    # re-estimation, which rewrites the tree from the reference first.  A reader's exception surfaces where its result is taken.
    from concurrent.futures import ThreadPoolExecutor
    pool = ThreadPoolExecutor(max_workers=2)
    ahead = {}
    if not options.dist_fp and not options.database_fp and not (options.tree_fp and options.reestimate_backbone):
        if options.ref_fp:
            ahead['ref'] = pool.submit(read_alignment, options.ref_fp, options.protein_seqs, False)
        qpath = options.query_fp or options.extended_ref_fp
        if qpath:
            ahead['query'] = pool.submit(read_alignment, qpath, options.protein_seqs, options.mask_lowconfidence)
    if options.database_fp:  # run_apples.py:25-35,69-75: tree, extended Newick and reduced reference from the cache
        from apples_amd import database
        tree, newick, reference, db_protein, db_threshold = database.load(options.database_fp)
        if options.dist_fp:
            reference = None  # database sequences are ignored with -d (apples/OptionsRun.py:92-97)
        else:
            if db_protein != bool(options.protein_seqs):
                raise ValueError('the database was built %s -p, the run was started %s it'
                                 % (('with', 'without') if db_protein else ('without', 'with')))
            if db_threshold != options.filt_threshold:
                # the reference's reduced reference keeps the threshold it was built with and walks the
                # representatives with it (apples/Reference.py:82,146); -f at run time does not reach it
                logging.info('[%s] The database was built with -f %s: that threshold is used for the observed sets.'
                             % (time.strftime('%H:%M:%S'), db_threshold))
                options.filt_threshold = db_threshold
        logging.info('[%s] Tree and reduced reference are loaded from the APPLES database in %.3f seconds.'
                     % (time.strftime('%H:%M:%S'), time.time() - start))
    if options.tree_fp:  # the user's tree wins over the database's (run_apples.py:37-38)
        if options.reestimate_backbone:  # apples/prepareTree.py:20-21 (off with -D and with -d)
            if options.ref_fp:
                from apples_amd.reestimate import cleanup, reestimate_backbone
                reestimate_backbone(options)  # rewrites options.tree_fp
            else:
                cleanup = None
                logging.warning('Backbone branch lengths are used as given: reestimation needs the reference alignment (-s).')
        tree = read_tree(options.tree_fp)
        if options.reestimate_backbone and options.ref_fp and not options.debug_mode:
            cleanup(options)  # (--debug keeps the resolved tree, FastTree's log and its answer)
        newick = pool.submit(extended_newick, tree)  # (needed when the output is written: formed beside the placement)
        logging.info('[%s] Tree is parsed and preprocessed in %.3f seconds.' % (time.strftime('%H:%M:%S'), time.time() - start))

    ngpu = options.num_gpus
    if ngpu <= 0:
        import ctypes
        from apples_amd.engine import load_library
        load_library()
        hip = ctypes.CDLL('libamdhip64.so')
        n = ctypes.c_int(0)
        hip.hipGetDeviceCount(ctypes.byref(n))
        ngpu = max(n.value, 1)
    devices = list(range(ngpu))

    if options.dist_fp:
        names, cols, D = read_distance_table(options.dist_fp)  # text (native scanner) or .npz
        worker = QueryWorker(tree, options, None, devices)
        startq = time.time()
        out_names, rows = worker.run_distances(names, cols, D, rows='arrays')
    else:
        start = time.time()
        ref = reference.aln if reference is not None else \
            (ahead['ref'].result() if 'ref' in ahead else read_alignment(options.ref_fp, options.protein_seqs, False))  # reference rows are never masked
        if reference is not None:
            clusters = None
        elif options.clusters_fp:
            clusters = read_treecluster(options.clusters_fp)
        elif options.no_clusters:
            clusters = None
            if any(n not in tree.name_to_node for n in ref.names):
                # only backbone leaves are ever observed (the reference's clusters name tree leaves)
                clusters = [('-1', [n for n in ref.names if n in tree.name_to_node])]
        else:  # as the reference: max-diameter clusters at 1.2 x the filter threshold (Reference.py:87)
            from apples_amd import treecluster
            clusters = treecluster.grouped(tree, options.filt_threshold * 1.2)
            in_aln = set(ref.names)
            missing = [n for _, g in clusters for n in g if n not in in_aln]
            if missing:
                raise KeyError('backbone leaf %s has no sequence in the reference alignment' % missing[0])
        if reference is None:
            reference = ReducedReference(ref, options.protein_seqs, clusters)
        logging.info('[%s] Reduced reference is prepared in %.3f seconds.' % (time.strftime('%H:%M:%S'), time.time() - start))
        if options.query_fp:
            q = ahead['query'].result() if 'query' in ahead else read_alignment(options.query_fp, options.protein_seqs, options.mask_lowconfidence)
            if len(q) and q.length != ref.length:
                raise ValueError('the query alignment has %d sites, the reference alignment %d' % (q.length, ref.length))
            qnames, qseqs = q.names, q.seqs
        else:
            ext = ahead['query'].result() if 'query' in ahead else read_alignment(options.extended_ref_fp, options.protein_seqs, options.mask_lowconfidence)
            if ext.length != ref.length:
                raise ValueError('the extended alignment has %d sites, the reference alignment %d' % (ext.length, ref.length))
            keep = [i for i, n in enumerate(ext.names) if n not in ref.index]
            qnames, qseqs = [ext.names[i] for i in keep], ext.seqs[keep]
        worker = QueryWorker(tree, options, reference, devices)
        startq = time.time()
        out_names, rows = worker.run_sequences(qnames, qseqs, rows='arrays')
    logging.info('[%s] Processed all queries in %.3f seconds.' % (time.strftime('%H:%M:%S'), time.time() - startq))
    worker.close()

    # join_jplace + json.dumps(sort_keys=True, indent=4) of the reference (run_apples.py:106-118), streamed
    invocation = sys.argv if argv is None else ['run_apples.py'] + list(argv)
    if not isinstance(newick, str):
        newick = newick.result()
    pool.shutdown(wait=False)
    fb = open(options.output_fp, 'wb') if options.output_fp else sys.stdout.buffer
    if not write_native(fb, out_names, rows, newick, invocation):
        # a name json.dumps would escape, or no native library: the same text from Python
        from apples_amd.engine import F_EXACT, F_INSUFFICIENT, F_DEGENERATE  # noqa: F401
        prow = [[int(e), 0, 1, 0, 0] if k == 1 else [int(e), float(er), 1, float(d), 0 if k == 2 else float(pe)]
                for e, er, d, pe, k in zip(rows['edge'].tolist(), rows['error'].tolist(), rows['distal'].tolist(),
                                           rows['pendant'].tolist(), rows['kind'].tolist())]
        keep = keep_mask([r[0] for r in prow])
        for piece in iter_text(((n, r) for n, r, k in zip(out_names, prow, keep) if k), newick, invocation):
            fb.write(piece.encode())
    if options.output_fp:
        fb.close()
    else:
        fb.flush()
    logging.warning('[%s] APPLES finished in %.3f seconds.' % (time.strftime('%H:%M:%S'), time.time() - startb))


if __name__ == '__main__':
    main()
