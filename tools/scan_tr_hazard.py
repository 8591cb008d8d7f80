"""Developer check on the ISA hipcc emits: an MFMA that reads a register as SrcA / SrcB, followed within three
instructions by a ds_read_b64_tr_b16 that redefines it -- the schedule that gave wrong sums in the weight-gradient kernel.

    hipcc ... -S --cuda-device-only file.hip -o file.s ; python tools/scan_tr_hazard.py file.s"""
import re
import sys


def regs(tok):
    m = re.match(r'v\[(\d+):(\d+)\]', tok)
    if m:
        return set(range(int(m.group(1)), int(m.group(2)) + 1))
    m = re.match(r'v(\d+)$', tok)
    return {int(m.group(1))} if m else set()


def scan(path, window=3):
    lines = open(path).read().split('\n')
    kern, hits, recent = None, {}, []
    for i, l in enumerate(lines):
        t = l.strip()
        m = re.match(r'^(_Z[\w.$]+):', t)
        if m:
            kern, recent = m.group(1), []
        if t.startswith('v_mfma'):
            ops = [o.strip() for o in t.split(None, 1)[1].split(',')]
            recent = (recent + [(i, regs(ops[1]) | regs(ops[2]))])[-4:]
        elif t.startswith('ds_read_b64_tr_b16'):
            d = regs(t.split(None, 1)[1].split(',')[0].strip())
            for j, src in recent:
                n = sum(1 for k in range(j + 1, i) if lines[k].strip() and not lines[k].strip().startswith((';', '.')))
                if d & src and n <= window:
                    hits[kern] = hits.get(kern, 0) + 1
    return hits


def scan_raw(path, window=2):
    """The other half of the schedule: a ds_read_b64_tr_b16 whose result an MFMA reads with at most `window` instructions
    (waits not counted) in between -- the consumer sits right behind the wait that covers the read."""
    lines = open(path).read().split('\n')
    kern, hits, recent = None, {}, []
    for i, l in enumerate(lines):
        t = l.strip()
        m = re.match(r'^(_Z[\w.$]+):', t)
        if m:
            kern, recent = m.group(1), []
        if t.startswith('ds_read_b64_tr_b16'):
            recent = (recent + [(i, regs(t.split(None, 1)[1].split(',')[0].strip()))])[-8:]
        elif t.startswith('v_mfma'):
            ops = [o.strip() for o in t.split(None, 1)[1].split(',')]
            src = regs(ops[1]) | regs(ops[2])
            for j, d in recent:
                n = sum(1 for k in range(j + 1, i) if lines[k].strip() and not lines[k].strip().startswith((';', '.', 's_waitcnt', 'ds_read')))
                if d & src and n <= window:
                    hits[kern] = hits.get(kern, 0) + 1
                    break
    return hits


def _issue_cycles(op):
    """Lower bound of the issue cycles an instruction occupies: s_nop k idles k + 1 cycles, anything else at least one."""
    m = re.match(r's_nop\s+(\d+)', op)
    return int(m.group(1)) + 1 if m else 1


def gate(path, war_cycles=8):
    """The two rules the shipped ISA is held to (tests/test_abi.py), per kernel {name: [(rule, line number, text)]}:
      WAR  an MFMA reads a register as SrcA / SrcB and a ds_read_b64_tr_b16 redefines it fewer than `war_cycles` issue
           cycles later (the failing weight-gradient schedule re-used a fragment register right behind its MFMA);
      RAW  an MFMA reads the result of a ds_read_b64_tr_b16 and no FULL `s_waitcnt lgkmcnt(0)` stands between the two
           (the failing schedule consumed its fragments behind counted waits, lgkmcnt(6) / lgkmcnt(2); the passing one
           only behind a drain).
    Neither sequence reproduces the wrong sums in isolation (tools/tr_hazard_probe.hip, tools/tr_lgkm_probe.hip); they
    are what distinguishes the failing build from the passing one, so no compiler update may bring them back unseen."""
    lines = open(path).read().split('\n')
    kern, hits = None, {}
    mf = []                      # recent MFMAs: (line index, source registers)
    pending = []                 # transposed reads not yet behind a full drain: (line index, destination registers)
    for i, l in enumerate(lines):
        t = l.strip()
        m = re.match(r'^(_Z[\w.$]+):', t)
        if m:
            kern, mf, pending = m.group(1), [], []
            continue
        if not t or t.startswith((';', '.')):
            continue
        op = t.split()[0]
        if op.startswith('v_mfma'):
            ops = [o.strip() for o in t.split(None, 1)[1].split(',')]
            src = regs(ops[1]) | regs(ops[2])
            for j, d in pending:
                if d & src:
                    hits.setdefault(kern, []).append(('RAW', i + 1, t))
                    break
            mf = (mf + [(i, src)])[-8:]
        elif op == 'ds_read_b64_tr_b16':
            d = regs(t.split(None, 1)[1].split(',')[0].strip())
            for j, src in mf:
                if d & src:
                    cyc = sum(_issue_cycles(lines[k].strip()) for k in range(j + 1, i)
                              if lines[k].strip() and not lines[k].strip().startswith((';', '.')))
                    if cyc < war_cycles:
                        hits.setdefault(kern, []).append(('WAR', i + 1, t))
                        break
            pending.append((i, d))
        elif op == 's_waitcnt' and re.search(r'lgkmcnt\(0\)', t):
            pending = []
    return hits


if __name__ == '__main__':
    if '--gate' in sys.argv:
        bad = 0
        for p in [a for a in sys.argv[1:] if not a.startswith('--')]:
            h = gate(p)
            bad += len(h)
            print(p, 'kernels violating the gate:', len(h))
            for k, v in h.items():
                print(f'  {len(v):4d}  {k[:100]}  first: {v[0]}')
        sys.exit(1 if bad else 0)
    for p in sys.argv[1:]:
        h = scan(p)
        print(p, 'kernels with the pattern:', len(h))
        for k, v in sorted(h.items(), key=lambda kv: -kv[1])[:40]:
            print(f'  {v:4d}  {str(k)[:110]}')
        h = scan_raw(p)
        print(p, 'kernels whose MFMAs read a transposed fragment right behind its wait:', len(h))
        for k, v in sorted(h.items(), key=lambda kv: -kv[1])[:40]:
            print(f'  {v:4d}  {str(k)[:110]}')
