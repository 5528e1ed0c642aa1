import sys
sys.path.insert(0, 'tests'); sys.path.insert(0, '.')
import conftest
import numpy as np
import test_bb_traces as tb
from helpers import make_controller
tr = np.load('tests/golden/bb_traces.npz', allow_pickle=False)
for name in tb.CASES:
    T, nub, terminal = (int(v) for v in tr[name + '_meta'])
    ctrl = make_controller(str(tr[name + '_fixture']), T=T, terminal=bool(terminal), backend='hip')
    rule = tb.RULES[str(tr[name + '_rule'])]
    ws = None
    for s in range(3):
        key = '%s_s%d_' % (name, s)
        if key + 'order' not in tr.files: break
        x = tr[key + 'x0']
        sol, leaves, solves, _ = ctrl.feedforward(x, search_rule=rule, warm_start=ws, printing_period=None, frontier_width=1)
        line = '%s step %d: solves %d (trace %d), leaves %d (trace %d)' % (name, s, solves, int(tr[key + 'solves']), len(leaves), len(tr[key + 'leaves_lb']))
        if key + 'ws_fix' not in tr.files:
            print(line); break
        u0, e0 = tr[key + 'u0'], tr[key + 'e0']
        nuc = ctrl.mld.nu - ctrl.mld.nub
        ub = np.concatenate(sol.variables['ub']).round().astype(np.int8)
        if not np.array_equal(ub, tr[key + 'incumbent_fix']):
            print(line, 'incumbent differs (tie)'); break
        ws, _, _ = ctrl.construct_warm_start(leaves, x, u0[:nuc], u0[nuc:], e0)
        has_dual = np.array([n.extra.dual is not None for n in ws])
        same_fix = len(ws) == len(tr[key + 'ws_lb']) and np.array_equal(tb._fix_rows(ctrl, ws), tr[key + 'ws_fix'])
        print(line, 'cover %d (trace %d) same identifiers %s reopen flags equal %s' % (len(ws), len(tr[key + 'ws_lb']), same_fix,
              (np.mean(has_dual == tr[key + 'ws_has_dual']) if same_fix else 'n/a')))
