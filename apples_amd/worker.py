"""Batch form of PoolQueryWorker.runquery (apples/PoolQueryWorker.py:28-141): what
``pool.starmap(queryworker.runquery, queries)`` returns (run_apples.py:94-102), computed by the
HIP engine on one or more MI355X devices."""
import logging
import sys
import threading

import numpy as np

from .engine import (Engine, placement_rows, F_EXACT, F_INSUFFICIENT, F_MISPLACED, F_ZERO_NOT_IN_TREE, F_DEGENERATE,
                     F_PENDANT_INT)


def _shards(n, parts):
    """Contiguous blocks, rank r gets [r*n/parts, (r+1)*n/parts) -- input order is preserved by
    construction, as starmap does."""
    return [(r * n // parts, (r + 1) * n // parts) for r in range(parts)]


class QueryWorker:
    """State injected once (PoolQueryWorker.set_class_attributes, PoolQueryWorker.py:17-24)."""

    def __init__(self, tree, options, reference=None, devices=(0,)):
        # A one-shot run keeps its device batch buffers small: beyond a few tens of GiB the allocation itself takes seconds
        # (the driver hands out scrubbed memory: 2 - 4 s for the 96 GiB a resident context would take at 200 000 references, every
        # run, against 0.1 s for 24 GiB), which no batch size wins back on one pass over the queries (scripts/r04_cli_batch_exp.sh).
        # A cap of this worker's contexts (apples_params.batch_gib: min(cap, what free memory allows)), not of the process
        self.batch_gib = 24
        self.tree = tree
        self.options = options
        self.reference = reference
        self.devices = list(devices)
        self._engines = {}

    def _engine(self, k):
        """The context of shard k (one per entry of ``devices``; two entries may name the same device)."""
        device = self.devices[k]
        if k not in self._engines:
            o = self.options
            kw = dict(protein=o.protein_seqs, method=o.method_name, criterion=o.criterion_name,
                      negative=bool(o.negative_branch), threshold=o.filt_threshold,
                      baseobs=o.base_observation_threshold, overlap=o.minimum_alignment_overlap, device=device,
                      batch_gib=self.batch_gib)
            if self.reference is not None:
                aln = self.reference.eng_aln  # the clustered rows (every row, unless -s holds more than the tree)
                nodes = np.array([self.tree.name_to_node.get(n, -1) for n in aln.names], np.int32)
                self._engines[k] = Engine(self.tree, aln.seqs, nodes, clusters=self.reference.cluster_arrays(), **kw)
            else:
                self._engines[k] = Engine(self.tree, None, **kw)
        return self._engines[k]

    def close(self):
        for e in self._engines.values():
            e.close()
        self._engines = {}

    def _run_sharded(self, n, fn):
        """fn(engine, lo, hi) -> placement array for queries [lo, hi); one host thread per device."""
        parts = _shards(n, len(self.devices))
        outs = [None] * len(parts)
        errs = []

        def work(k):
            try:
                lo, hi = parts[k]
                outs[k] = fn(self._engine(k), lo, hi)
            except Exception as e:  # an exception in any worker aborts the run, as starmap does
                errs.append(e)

        if len(parts) == 1:
            work(0)
        else:
            th = [threading.Thread(target=work, args=(k,)) for k in range(len(parts))]
            [t.start() for t in th]
            [t.join() for t in th]
        if errs:
            raise errs[0]
        return np.concatenate(outs) if outs else np.zeros(0)

    # ------------------------------------------------------------------ alignment input
    def run_sequences(self, names, seqs, rows=False):
        aln = self.reference.eng_aln
        # the entry runquery deletes: query name is a backbone leaf and names a reference row (:63-66)
        self_rows = np.array([aln.index.get(n, -1) if n in self.tree.name_to_node else -1 for n in names], np.int32)
        out = self._run_sharded(len(names), lambda eng, lo, hi: eng.place_sequences(seqs[lo:hi], self_rows[lo:hi]))
        return self._rows(names, out, arrays=rows == 'arrays') if rows else self._to_jplace(names, out)

    # ------------------------------------------------------------------ distance-table input
    def run_distances(self, names, cols, D, rows=False):
        col_nodes = np.array([self.tree.name_to_node.get(c, -1) for c in cols], np.int32)
        col_index = {c: i for i, c in enumerate(cols)}
        self_cols = np.array([col_index.get(n, -1) if n in self.tree.name_to_node else -1 for n in names], np.int32)
        out = self._run_sharded(len(names), lambda eng, lo, hi: eng.place_distances(D[lo:hi], col_nodes, self_cols[lo:hi]))
        return self._rows(names, out, arrays=rows == 'arrays') if rows else self._to_jplace(names, out)

    # ------------------------------------------------------------------ result assembly
    def _rows(self, names, out, arrays=False):
        """(names as printed, jplace p rows): the branches of runquery that are visible in the output
        (PoolQueryWorker.py:63-70,83-88,119-130), flags first so that the common case stays vectorised."""
        flags = out['flags']
        zero_bad = np.nonzero(flags & F_ZERO_NOT_IN_TREE)[0]
        degenerate = np.nonzero(flags & F_DEGENERATE)[0]
        in_tree = self.tree.name_to_node
        names = list(names)
        for i, name in enumerate(names):
            if name in in_tree:
                logging.warning('The query named %s exists in the backbone. Changing its name to %s-query.' % (name, name))
                names[i] = name + '-query'
        # errors in input order, as the reference's starmap would raise the first one
        if len(zero_bad) or len(degenerate):
            i = min([int(x[0]) for x in (zero_bad, degenerate) if len(x)])
            if flags[i] & F_ZERO_NOT_IN_TREE:
                raise KeyError('query %s has a zero distance to a reference that is not a leaf of the backbone tree' % names[i])
            raise ValueError('query %s: fewer than two of its observed references are leaves of the backbone tree' % names[i])
        rows = None if arrays else placement_rows(out)
        edge = out['edge'].copy() if arrays else None
        for i in np.nonzero(flags & F_INSUFFICIENT)[0]:
            sys.stderr.write('Taxon {} cannot be placed. At least three non-infinity distances '
                             'should be observed to place a taxon. '
                             'Consequently, this taxon is ignored (no output).\n'.format(names[i]))
        mis = np.nonzero(((flags & F_MISPLACED) != 0) & ((flags & (F_EXACT | F_INSUFFICIENT)) == 0))[0]
        for i in mis:
            ignored = ''
            if self.options.exclude_intplace:
                if arrays:
                    edge[i] = -1
                else:
                    rows[i][0] = -1
                ignored = ' Consequently, this sequence is ignored (no output).'
            logging.warning('Best placement for query sequence %s has zero pendant edge length and placed at an '
                            'internal node with a non-zero least squares error. This is a potential misplacement.%s'
                            % (names[i], ignored))
        if arrays:
            # the p rows as columns (what apples_amd/jplace.py's native writer takes): kind 1 = [edge, 0, 1, 0, 0], 2 = the
            # pendant is the clamped int 0, 0 = five numbers as they are (engine.placement_rows)
            kind = np.where((flags & (F_EXACT | F_INSUFFICIENT | F_DEGENERATE)) != 0, 1,
                            np.where((flags & F_PENDANT_INT) != 0, 2, 0)).astype(np.uint8)
            return names, {'edge': edge, 'error': np.ascontiguousarray(out['error']), 'distal': np.ascontiguousarray(out['distal']),
                           'pendant': np.ascontiguousarray(out['pendant']), 'kind': kind}
        return names, rows

    def _to_jplace(self, names, out):
        """One jplace dict per query, as runquery returns (PoolQueryWorker.py:36-37)."""
        names, rows = self._rows(names, out)
        return [{'placements': [{'p': [row], 'n': [name]}]} for name, row in zip(names, rows)]
