"""Brute-force bank-conflict check of the wave-private LDS tile image used by
csrc/edge_mfma.hip (DESIGN.md "LDS tile image").  Banking rules from
/opt/skills/guides/MI355X_MICROARCH.md section LDS:
  ds_write_b128: 8 groups of 8 contiguous lanes, 32 banks
  ds_read_b128 : 4 groups of 16 lanes (listed below), 64 banks
  ds_read_b32  : 2 groups of 32 lanes, 32 banks
An access of w dwords at dword address a occupies banks a..a+w-1 (mod nb).
cost(group) = max over banks of the number of DISTINCT addresses on it."""
import itertools, sys

B128_GROUPS = [
    list(range(0, 4)) + list(range(12, 16)) + list(range(20, 28)),
    list(range(4, 12)) + list(range(16, 20)) + list(range(28, 32)),
    list(range(32, 36)) + list(range(44, 48)) + list(range(52, 60)),
    list(range(36, 44)) + list(range(48, 52)) + list(range(60, 64)),
]
B32_GROUPS = [list(range(0, 32)), list(range(32, 64))]
W128_GROUPS = [list(range(8 * g, 8 * g + 8)) for g in range(8)]


def cost(groups, addr_of_lane, width, nbanks):
    worst = 0
    for g in groups:
        banks = {}
        for l in g:
            a = addr_of_lane(l)
            if a is None:
                continue
            for w in range(width):
                banks.setdefault((a + w) % nbanks, set()).add(a)
        worst = max(worst, max((len(v) for v in banks.values()), default=0))
    return worst


def make_addr(DH, stride, f):
    nchunk = DH // 4
    def addr(j, c):
        return j * stride + (((c >> 2) ^ f(j)) % nchunk if False else ((c >> 2) ^ f(j))) * 4 + (c & 3)
    return addr


def evaluate(DH, stride, f, L=20, verbose=False):
    addr = make_addr(DH, stride, f)
    lanes_per_row = DH // 4
    res = {}
    # global-load lane map: lane l -> row l // lanes_per_row (+ rows_per_instr * t), chunk l % lanes_per_row
    rpi = 64 // lanes_per_row
    worst = 0
    for t in range((L + rpi - 1) // rpi):
        def a(l, t=t):
            j = l // lanes_per_row + rpi * t
            return addr(j, 4 * (l % lanes_per_row)) if j < L else None
        worst = max(worst, cost(W128_GROUPS, a, 4, 32))
    res['write_b128'] = worst
    # row operand: lane (m = l & 15, ks = l >> 4); tile mt=0: j = m; mt=1 (quarter): j = 16 + m // 4 if m % 4 == 0
    kpl = DH // 4                    # channels per lane (k-steps)
    worst = 0
    for mt in range(2):
        for half in range(max(1, kpl // 4)):
            def a(l, mt=mt, half=half):
                m, ks = l & 15, l >> 4
                if mt == 0:
                    j = m
                else:
                    if m % 4:
                        return None
                    j = 16 + m // 4
                if j >= L:
                    return None
                return addr(j, kpl * ks + 4 * half)
            w = 4 if kpl >= 4 else kpl
            worst = max(worst, cost(B128_GROUPS, a, 4, 64))
    res['rowop_b128'] = worst
    # column operand: lane (c' = l & 15, ks = l >> 4), k-step q: j = 4 ks + q (q<4), j = 16 + ks (q=4)
    worst = 0
    for mc in range(DH // 16):
        for q in range(5):
            def a(l, mc=mc, q=q):
                c, ks = (l & 15) + 16 * mc, l >> 4
                j = 4 * ks + q if q < 4 else 16 + ks
                return addr(j, c) if j < L else None
            worst = max(worst, cost(B32_GROUPS, a, 1, 32))
    res['colop_b32'] = worst
    return res


if __name__ == '__main__':
    for DH in (32, 16):
        best = None
        nchunk = DH // 4
        cands = []
        for stride in range(DH, DH + 17, 4):
            for name, f in [('none', lambda j: 0),
                            ('f1', lambda j: ((((j >> 2) & 1) << 2) | ((j >> 1) & 3)) % (DH // 4)),
                            ('f1x', lambda j: (((((j >> 2) ^ (j >> 3)) & 1) << 2) | ((j >> 1) & 3)) % (DH // 4)),   # swz<DH> since round 2
                            ('f2', lambda j: (j >> 1) % (DH // 4)),
                            ('f3', lambda j: (j >> 2) % (DH // 4)),
                            ('f4', lambda j: j % (DH // 4)),
                            ('f5', lambda j: ((j >> 1) ^ (j >> 3)) % (DH // 4))]:
                r = evaluate(DH, stride, f)
                cands.append((sum(r.values()), stride, name, r))
        cands.sort(key=lambda t: (t[0], t[1]))
        print('DH', DH)
        for c in cands[:6]:
            print('  ', c)
