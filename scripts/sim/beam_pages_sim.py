"""Lane-level emulation (numpy, 64 lanes) of beam_pop_hybrid / beam_push_hybrid with the round-3 paged spill layout
(device_search.h: beam_spill_off, pages of 64 x 12 B for heap levels 8..12, heap-ordered tail beyond), checked against
libstdc++'s pop_heap / push_heap restated here.  Run on the CPU before the kernel goes to a GPU: it pins the index
arithmetic (bijection of the placement, page lanes of the path, left-only children, tail windows)."""
import numpy as np

KL, PAGED, PD, TAIL = 255, 8191, 192, 128 * 192


def clz32(x):
    return 32 - int(x).bit_length()


def off(i):
    hp = i + 1
    l = (31 - clz32(hp)) - 7
    if hp < PAGED + 1:
        r = (hp & ((1 << l) - 1)) | (1 << l)
        return ((hp >> l) - 128) * PD + 3 * r
    return TAIL + 3 * (hp - (PAGED + 1))


class Heap:
    def __init__(self, cap):
        self.lds = np.zeros((KL + 1, 2))        # key, id
        self.g = np.full(TAIL + 3 * max(0, cap - PAGED) + 4, -7.0)
        self.touched = set()

    def raw(self, i):
        if i < KL:
            return tuple(self.lds[i])
        o = off(i)
        return (self.g[o], self.g[o + 2])

    def put(self, i, e):
        if i < KL:
            self.lds[i] = e
        else:
            o = off(i)
            self.g[o] = e[0]; self.g[o + 2] = e[1]

    def key(self, i):
        return self.raw(i)[0]


def before(a, b):      # std::greater on est: min-heap
    return a > b


def pop_hybrid(h, size):
    length = size - 1
    v = h.raw(length)
    hp, d = 1, 0
    for _ in range(7):
        hole = hp - 1
        left = before(h.key(2 * hole + 2), h.key(2 * hole + 1))
        hp = 2 * hp + 1 - int(left)
        d += 1
    lanes = np.arange(64)
    r = lanes + 2
    dr = np.array([31 - clz32(x) for x in r])
    w = [None] * 64

    def walk(keys, idx):
        nonlocal hp, d
        E2 = [(l & 1) == 0 and l < 62 and idx[l] + 1 < length for l in range(64)]
        M = [E2[l] and before(keys[l + 1], keys[l]) for l in range(63)] + [False]
        P, steps = 1, 0
        while steps < 5:
            bit = 2 * P - 2
            if not E2[bit]:
                break
            P = 2 * P + 1 - int(M[bit])
            steps += 1
        hp = (hp << steps) | (P & ((1 << steps) - 1))
        d += steps
        return steps

    steps = 0
    if 2 * hp - 1 < length:
        idx = ((hp << dr) | (r & ((1 << dr) - 1))) - 1
        keys = np.zeros(64)
        for l in range(62):
            if idx[l] < length:
                o = (hp - 128) * PD + 3 * r[l]
                assert o == off(idx[l]), (o, off(idx[l]), idx[l])
                w[l] = (h.g[o], h.g[o + 2])
                keys[l] = w[l][0]
        steps = walk(keys, idx)
    while steps == 5 and 2 * hp < length:
        idx = ((hp << dr) | (r & ((1 << dr) - 1))) - 1
        keys = np.zeros(64)
        for l in range(62):
            if idx[l] < length:
                keys[l] = h.g[off(idx[l])]
        steps = walk(keys, idx)
    if (length & 1) == 0 and hp - 1 == (length - 2) >> 1:
        hp = 2 * hp
        d += 1
    hole = hp - 1
    es, cs = [None] * 64, [False] * 64
    dsts = [None] * 64
    for lane in range(64):
        t = lane if lane < d else 0
        my_dst = (hp >> (d - t)) - 1
        src_hp = hp >> (d - t - (1 if lane < d else 0))
        my_src = src_hp - 1
        dsts[lane] = my_dst
        if lane < d:
            if t < 7:
                e = h.raw(my_src)
            elif t <= 11:
                lvl = t - 6
                pl = ((src_hp & ((1 << lvl) - 1)) | (1 << lvl)) - 2
                e = w[pl]
                assert e is not None and e == h.raw(my_src), (t, pl, my_src)
            else:
                e = h.raw(my_src)
            es[lane] = e
            cs[lane] = before(e[0], v[0])
    stay = [lane for lane in range(d) if not cs[lane]]
    fin = (max(stay) + 1) if stay else 0
    pbase = ((hp >> (d - 7)) - 128) * PD

    def page_rel(node_hp, lvl):
        return (node_hp & ((1 << lvl) - 1)) | (1 << lvl)

    def put_level(level, node_hp, val):      # the kernel's specialised addressing must agree with the placement function
        if level <= 7:
            assert node_hp - 1 < KL
        elif level <= 12:
            assert pbase + 3 * page_rel(node_hp, level - 7) == off(node_hp - 1), (level, node_hp)
        else:
            assert TAIL + 3 * (node_hp - (PAGED + 1)) == off(node_hp - 1)
        h.put(node_hp - 1, val)

    for lane in range(fin):
        put_level(lane, hp >> (d - lane), es[lane])
    if fin == d:
        put_level(d, hp, v)
    else:
        put_level(fin, hp >> (d - fin), v)


def push_hybrid(h, hole, v):
    hp = hole + 1
    depth = 31 - clz32(hp)
    if KL <= hole and hp <= PAGED:           # the kernel's fast path: every HBM ancestor in the leaf's own page
        lvl = depth - 7
        pbase = ((hp >> lvl) - 128) * PD
        rl = (hp & ((1 << lvl) - 1)) | (1 << lvl)
        for s in range(0, depth + 1):
            if s < lvl:
                assert pbase + 3 * (rl >> s) == off((hp >> s) - 1), (hole, s)
            else:
                assert (hp >> s) - 1 < KL
    if hole >= KL:                           # the touch of the expansion head: ancestors of levels 8..12 below the leaf's level
        pb = ((hp >> (depth - 7)) - 128) * PD
        for j in range(1, 6):
            if 7 + j < depth:
                sh = depth - 7 - j
                rj = ((hp >> sh) & ((1 << j) - 1)) | (1 << j)
                assert pb + 3 * rj == off((hp >> sh) - 1), (hole, j)
    es, down = {}, {}
    for t in range(1, depth + 1):
        pt = (hp >> t) - 1
        es[t] = h.raw(pt)
        down[t] = before(es[t][0], v[0])
    m = 0
    while m < depth and down[m + 1]:
        m += 1
    for t in range(1, m + 1):
        h.put((hp >> (t - 1)) - 1, es[t])
    h.put((hp >> m) - 1, v)


# ---- libstdc++ -------------------------------------------------------------------------------------
def std_push(a, v):
    a.append(v)
    hole = len(a) - 1
    while hole > 0:
        p = (hole - 1) >> 1
        if not before(a[p][0], v[0]):
            break
        a[hole] = a[p]
        hole = p
    a[hole] = v


def std_pop(a):
    v = a[-1]
    a[-1] = a[0]
    length = len(a) - 1
    hole = 0
    child = 0
    while child < (length - 1) // 2:
        child = 2 * (child + 1)
        if before(a[child][0], a[child - 1][0]):
            child -= 1
        a[hole] = a[child]
        hole = child
    if (length & 1) == 0 and child == (length - 2) // 2:
        child = 2 * (child + 1)
        a[hole] = a[child - 1]
        hole = child - 1
    while hole > 0:
        p = (hole - 1) >> 1
        if not before(a[p][0], v[0]):
            break
        a[hole] = a[p]
        hole = p
    a[hole] = v
    a.pop()


def run(n_fill, distinct, seed):
    rng = np.random.default_rng(seed)
    h = Heap(n_fill + 2000)
    ref = []
    size = 0
    nid = 0

    def push():
        nonlocal size, nid
        v = (float(rng.integers(0, distinct)), float(nid))
        nid += 1
        std_push(ref, v)
        push_hybrid(h, size, v)     # (the LDS-only routines are the same algorithm with every index below 255)
        size += 1

    def pop():
        nonlocal size
        if size == 0:
            return
        std_pop(ref)
        if size > 1:
            if size <= KL:
                # heap_pop_wave's logic == pop_hybrid's with no HBM part; emulate with the std algorithm on the view
                a = [h.raw(i) for i in range(size)]
                std_pop(a)
                for i, e in enumerate(a):
                    h.put(i, e)
            else:
                pop_hybrid(h, size)
        size -= 1

    for _ in range(n_fill):
        push()
    for op in rng.integers(0, 2, 400):
        push() if op else pop()
    for _ in range(min(n_fill, 300)):
        pop()
    for _ in range(200):
        push()
    got = [h.raw(i) for i in range(size)]
    assert got == ref, (n_fill, distinct)


if __name__ == "__main__":
    offs = [off(i) for i in range(KL, 70000)]
    assert len(set(offs)) == len(offs) and min(offs) >= 6 and max(offs) == TAIL + 3 * (70000 - (PAGED + 1)), (min(offs), max(offs))
    assert max(o for o, i in zip(offs, range(KL, 70000)) if i < PAGED) < TAIL
    for n_fill in (254, 255, 256, 257, 300, 511, 512, 513, 1023, 1024, 4095, 4096, 4097, 8190, 8191, 8192, 8193, 9000, 16383, 16384, 16500, 40000):
        for distinct in (8, 1 << 20):
            run(n_fill, distinct, n_fill * 3 + distinct)
        print("ok", n_fill, flush=True)
