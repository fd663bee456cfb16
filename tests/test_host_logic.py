"""Host logic of the C-ABI library on CPU (no GPU, no HIP call): v2 index reader/writer, the
reference-layout -> device-layout repacker, and the host mirror of the query encoder."""
import ctypes as C

import numpy as np
import pytest

from golden_util import DATASETS, fixture_path


def _lib():
    from cphnsw_mi355x import _lib
    return _lib


@pytest.mark.parametrize("name,bits", [("g128", 1), ("g128", 4), ("sift96", 4), ("g16", 2), ("g1024", 2)])
def test_index_reader_writer_is_byte_identical(tmp_path, name, bits):
    L = _lib()
    src = fixture_path(name, bits)
    dst = str(tmp_path / "copy.idx")
    L.check(L.lib().cph_host_rewrite_index(src.encode(), dst.encode()))
    assert open(dst, "rb").read() == open(src, "rb").read()


def test_reader_rejects_bad_files(tmp_path):
    L = _lib()
    bad = tmp_path / "bad.idx"
    bad.write_bytes(b"\x01" * 300)
    with pytest.raises(RuntimeError, match="Invalid magic"):
        L.check(L.lib().cph_host_rewrite_index(str(bad).encode(), str(tmp_path / "o").encode()))
    data = open(fixture_path("g16", 1), "rb").read()
    (tmp_path / "trunc.idx").write_bytes(data[: len(data) // 2])
    with pytest.raises(RuntimeError, match="truncated"):
        L.check(L.lib().cph_host_rewrite_index(str(tmp_path / "trunc.idx").encode(), str(tmp_path / "o").encode()))
    corrupt = bytearray(data)
    # first neighbour id of vertex 0 -> out of range
    n, dim, D = DATASETS["g16"]["n"], DATASETS["g16"]["dim"], 16
    from golden_util import vertex_layout
    vb, nb_off, codes = vertex_layout(D, 1)
    base = 68 + 248 + 72 + dim * 4 + n * 4 + n * 4 + n * D * 4
    ids_off = base + nb_off + codes + 3 * 128 + 64
    corrupt[ids_off:ids_off + 4] = (n + 5).to_bytes(4, "little")
    (tmp_path / "corrupt.idx").write_bytes(bytes(corrupt))
    with pytest.raises(RuntimeError, match="out of range"):
        L.check(L.lib().cph_host_rewrite_index(str(tmp_path / "corrupt.idx").encode(), str(tmp_path / "o").encode()))


@pytest.mark.parametrize("D", [16, 32, 64, 128, 256, 1024, 2048])
@pytest.mark.parametrize("bits", [1, 2, 4])
def test_repacker_layout_and_roundtrip(oracle, D, bits):
    """Every code bit, aux value and id lands where DESIGN.md §3 says, and the inverse is exact."""
    L = _lib()
    lay = oracle.layout(D, bits)
    nb_bytes = lay[0] - lay[1]
    rng = np.random.default_rng(D * 8 + bits)
    ref = np.zeros(nb_bytes, np.uint8)
    planes = rng.integers(0, 256, (bits, max(D // 8, 2), 32), dtype=np.uint8)
    if D == 16:
        planes = planes[:, :2]
    plane_stride = (planes.shape[1] * 32 + 63) // 64 * 64
    for b in range(bits):
        ref[lay[2] + b * plane_stride: lay[2] + b * plane_stride + planes.shape[1] * 32] = planes[b].reshape(-1)
    nop = rng.standard_normal(32).astype(np.float32)
    ipqo = rng.standard_normal(32).astype(np.float32)
    ipcp = rng.standard_normal(32).astype(np.float32)
    pop = rng.integers(0, 60000, 32).astype(np.uint16)
    wpop = rng.integers(0, 60000, 32).astype(np.uint16)
    ids = rng.integers(0, 2 ** 31, 32).astype(np.uint32)
    count = 29
    ref[lay[3]:lay[3] + 128] = nop.view(np.uint8)
    ref[lay[4]:lay[4] + 128] = ipqo.view(np.uint8)
    ref[lay[5]:lay[5] + 128] = ipcp.view(np.uint8)
    ref[lay[6]:lay[6] + 64] = pop.view(np.uint8)
    if bits > 1:
        ref[lay[7]:lay[7] + 64] = wpop.view(np.uint8)
    ref[lay[8]:lay[8] + 128] = ids.view(np.uint8)
    ref[lay[9]:lay[9] + 4] = np.array([count], np.uint32).view(np.uint8)
    dev = np.zeros(1 << 17, np.uint8)
    back = np.zeros(nb_bytes, np.uint8)
    nbytes = C.c_uint64(0)
    L.check(L.lib().cph_host_repack_block(D, bits, ref.ctypes.data, dev.ctypes.data, C.byref(nbytes),
                                          back.ctypes.data))
    stride = nbytes.value
    PW = max(1, D // 32)
    T = bits * PW
    assert stride == (32 * T * 4 + 512 + 128 + 4 + 63) // 64 * 64
    dw = dev[:32 * T * 4].view(np.uint32)
    wide = D >= 128
    NH = 2 if (wide and T // 4 >= 2) else 1
    CPL = (T // 4 // NH) if wide else 0
    for b in range(bits):
        for w in range(PW):
            for i in (0, 7, 31):
                want = 0
                for s_ in range(4):
                    sp = 4 * w + s_
                    if sp < planes.shape[1]:
                        want |= int(planes[b, sp, i]) << (8 * s_)
                t = b * PW + w
                if wide:
                    ck, e = divmod(t, 4)
                    h, k = (divmod(ck, CPL) if NH == 2 else (0, ck))
                    idx = (k * NH * 32 + h * 32 + i) * 4 + e
                else:
                    idx = t * 32 + i
                assert int(dw[idx]) == want, (D, bits, b, w, i)
    aux = dev[32 * T * 4: 32 * T * 4 + 512].view(np.uint32).reshape(32, 4)
    assert np.array_equal(aux[:, 0], nop.view(np.uint32)) and np.array_equal(aux[:, 1], ipqo.view(np.uint32))
    assert np.array_equal(aux[:, 2], ipcp.view(np.uint32))
    assert np.array_equal(aux[:, 3] & 0xFFFF, pop.astype(np.uint32))
    if bits > 1:
        assert np.array_equal(aux[:, 3] >> 16, wpop.astype(np.uint32))
    dids = dev[32 * T * 4 + 512: 32 * T * 4 + 640].view(np.uint32)
    assert np.array_equal(dids[:count], ids[:count]) and (dids[count:] == 0xFFFFFFFF).all()
    # inverse: identical except that slots >= count now carry the invalid id
    expect = ref.copy()
    expect[lay[8] + 4 * count: lay[8] + 128] = 0xFF
    if bits == 1:
        pass
    assert np.array_equal(back, expect)


@pytest.mark.parametrize("D,dim", [(16, 10), (128, 128), (128, 96), (1024, 960), (2048, 1536)])
def test_host_query_encoder_matches_reference(gold, D, dim):
    L = _lib()
    q = gold[f"E/{D}/{dim}/q"]
    for i in range(len(q)):
        lut = np.zeros((D // 4, 16), np.uint8)
        co = np.zeros(3, np.float32)
        masks = np.zeros((max(1, D // 32), 4), np.uint32)
        qq = np.ascontiguousarray(q[i], np.float32)
        L.check(L.lib().cph_host_encode_query(dim, qq.ctypes.data, lut.ctypes.data, co.ctypes.data,
                                              masks.ctypes.data))
        assert np.array_equal(lut, gold[f"E/{D}/{dim}/lut"][i])
        assert co.tobytes() == gold[f"E/{D}/{dim}/coeffs"][i].tobytes()
        # masks are the bit-sliced scalars: bit t of masks[w][j] == bit j of q_u[32w+t]
        qu = gold[f"E/{D}/{dim}/lut"][i][:, [1, 2, 4, 8]].reshape(-1)  # lut[seg][1<<b] = q_u[4seg+b]
        for d in range(0, D, max(1, D // 16)):
            for j in range(4):
                assert ((int(masks[d // 32, j]) >> (d % 32)) & 1) == ((int(qu[d]) >> j) & 1)
