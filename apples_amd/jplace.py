"""jplace assembly and emission (apples/jutil.py:1-19, run_apples.py:106-118)."""
import json


def join_jplace(lst):
    """Concatenate per-query results, dropping unplaceable ones (edge -1) -- except that the first
    result is kept as is whenever there is more than one (apples/jutil.py:11-18)."""
    result = lst[0]
    if len(lst) == 1:
        if result['placements'][0]['p'][0][0] == -1:
            result['placements'] = []
    else:
        extra = [r['placements'][0] for r in lst[1:] if r['placements'][0]['p'][0][0] != -1]
        result['placements'] = result['placements'] + extra
    return result


def finish(result, tree_string, argv):
    result['tree'] = tree_string
    result['metadata'] = {'invocation': ' '.join(argv)}
    result['fields'] = ['edge_num', 'likelihood', 'like_weight_ratio', 'distal_length', 'pendant_length']
    result['version'] = 3
    return result


def dumps(result):
    return json.dumps(result, sort_keys=True, indent=4) + '\n'


# ---------------------------------------------------------------------------------------------
# Streaming writer.  json.dumps with indent runs the pure-Python encoder over one
# giant dict: 1.4 s per 100 000 placements, several times the whole GPU pass.  The text below is
# the same text, byte for byte, assembled directly.

def _num(x):
    if isinstance(x, float):
        r = float.__repr__(x)
        if r in ('nan', 'inf', '-inf'):  # json's allow_nan spellings
            return {'nan': 'NaN', 'inf': 'Infinity', '-inf': '-Infinity'}[r]
        return r
    return str(int(x))


def iter_text(placements, tree_string, argv):
    """Yield the jplace text in pieces; ``placements`` is an iterable of (name, p row), already
    filtered as :func:`join_jplace` would (see :func:`keep_mask`).  Same bytes as
    ``dumps(finish(join_jplace(results), tree_string, argv))``."""
    q = json.dumps
    yield ('{\n    "fields": [\n        "edge_num",\n        "likelihood",\n        "like_weight_ratio",\n'
           '        "distal_length",\n        "pendant_length"\n    ],\n    "metadata": {\n        "invocation": %s\n    },\n'
           % q(' '.join(argv)))
    first = True
    buf = []
    for name, row in placements:
        buf.append('%s\n        {\n            "n": [\n                %s\n            ],\n            "p": [\n'
                   '                [\n                    %s,\n                    %s,\n                    %s,\n'
                   '                    %s,\n                    %s\n                ]\n            ]\n        }'
                   % ('    "placements": [' if first else ',', q(name), _num(row[0]), _num(row[1]), _num(row[2]),
                      _num(row[3]), _num(row[4])))
        first = False
        if len(buf) >= 4096:
            yield ''.join(buf)
            buf = []
    if buf:
        yield ''.join(buf)
    yield '    "placements": [],\n' if first else '\n    ],\n'
    yield '    "tree": %s,\n    "version": 3\n}\n' % q(tree_string)


def keep_mask(edges):
    """Which results survive :func:`join_jplace`: unplaceable ones (edge -1) are dropped, except
    that the first result is kept as is whenever there is more than one (apples/jutil.py:11-18)."""
    keep = [e != -1 for e in edges]
    if len(keep) > 1:
        keep[0] = True
    return keep
