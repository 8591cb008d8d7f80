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


W64_GROUPS = [list(range(16 * g, 16 * g + 16)) for g in range(4)]     # ds_write_b64: 4 groups of 16 contiguous lanes, 32 banks


def check_proj():
    """Entry `proj`: the LDS images of csrc/proj_gemm.hip (byte addresses -> dword addresses).
    rows kernel, A planes of one 16-deep step: thread (c = t & 3, r0 = t >> 2) stores 8 bytes of row r0 at fragment slot
    32 h + (r ^ 4 h), half q (h = c >> 1, q = c & 1); a fragment read is lane (r, h) -> slot 32 h + (r ^ 4 h), 16 bytes.
    wgrad kernel: [16 rows][256 columns] bf16 rows of 512 bytes, byte (8 c) ^ ((row & 3) << 6); transposed reads: lane
    (h, gi, q, p) addresses row 8 h + q, byte ((64 tile) ^ (q << 6)) + 32 gi + 8 p (ds_read_b64_tr_b16 banks as ds_read_b64:
    two groups of 32 lanes, 64 banks)."""
    def wr(l):           # 64 consecutive threads of a wave: t = l (any wave: rows shift by 16, same residues)
        c, r0 = l & 3, l >> 2
        h, q = c >> 1, c & 1
        return (((32 * h + ((r0 & 31) ^ (4 * h))) << 4) + 8 * q) // 4
    def rd(l):
        r, h = l & 31, l >> 5
        return ((32 * h + (r ^ (4 * h))) << 4) // 4
    print('proj rows: A-plane ds_write_b64', cost(W64_GROUPS, wr, 2, 32), ' fragment ds_read_b128', cost(B128_GROUPS, rd, 4, 64))
    for row_bytes, ncol4 in ((512, 64), (256, 32)):
        worst_w = worst_r = 0
        for rbase in range(0, 16, 64 // ncol4 if ncol4 <= 64 else 1):
            def w(l, rbase=rbase):
                c, r = l % ncol4, rbase + l // ncol4
                return (r * row_bytes + ((8 * c) ^ ((r & 3) << 6))) // 4
            worst_w = max(worst_w, cost(W64_GROUPS, w, 2, 32))
        for tile in range(row_bytes // 64):
            for u in range(2):
                def r(l, tile=tile, u=u):
                    h, gi, q, p = l >> 5, (l >> 4) & 1, (l >> 2) & 3, l & 3
                    return ((8 * h + 4 * u + q) * row_bytes + (((tile & 3) ^ q) << 6) + ((tile >> 2) << 8) + 32 * gi + 8 * p) // 4
                worst_r = max(worst_r, cost(B32_GROUPS, r, 2, 64))
        print(f'proj wgrad ({row_bytes}-byte rows): plane ds_write_b64', worst_w, ' ds_read_b64_tr_b16', worst_r)


if __name__ == '__main__':
    if len(sys.argv) > 1 and sys.argv[1] == 'proj':
        check_proj()
        sys.exit(0)
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
