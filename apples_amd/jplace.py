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
