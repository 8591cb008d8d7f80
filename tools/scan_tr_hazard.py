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


if __name__ == '__main__':
    for p in sys.argv[1:]:
        h = scan(p)
        print(p, 'kernels with the pattern:', len(h))
        for k, v in sorted(h.items(), key=lambda kv: -kv[1])[:40]:
            print(f'  {v:4d}  {str(k)[:110]}')
        h = scan_raw(p)
        print(p, 'kernels whose MFMAs read a transposed fragment right behind its wait:', len(h))
        for k, v in sorted(h.items(), key=lambda kv: -kv[1])[:40]:
            print(f'  {v:4d}  {str(k)[:110]}')
