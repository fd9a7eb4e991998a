"""Derive the operand / scale lane mapping of the scaled fp8 MFMA (16x16x128) empirically. Prints the inferred maps."""
import ctypes, os, torch
here = os.path.dirname(os.path.abspath(__file__))
lib = ctypes.CDLL(os.path.join(here, "probe.so"))
lib.run_probe.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p]
dev = "cuda"
ONE = 0x38  # e4m3 1.0
st = torch.cuda.current_stream().cuda_stream
def run(A, B, sa, sb):
    n = max(A.shape[0], B.shape[0], sa.shape[0])
    D = torch.zeros(n, 64, 4, device=dev)
    rc = lib.run_probe(A.data_ptr(), 2048 if A.shape[0] > 1 else 0, B.data_ptr(), 2048 if B.shape[0] > 1 else 0, sa.data_ptr(),
                       64 if sa.shape[0] > 1 else 0, sb.data_ptr(), D.data_ptr(), n, st)
    torch.cuda.synchronize(); assert rc == 0
    # D[lane][reg] -> matrix [row = (lane>>4)*4+reg][col = lane&15]
    M = torch.zeros(n, 16, 16)
    Dc = D.cpu()
    for l in range(64):
        for r in range(4):
            M[:, (l >> 4) * 4 + r, l & 15] = Dc[:, l, r]
    return M
s127 = torch.full((1, 64), 127, dtype=torch.int32, device=dev)
ones = torch.full((1, 64, 32), ONE, dtype=torch.uint8, device=dev)
# P1: A one-hot scan (all 2048 positions), B all ones -> which output row lights up
A = torch.zeros(2048, 64, 32, dtype=torch.uint8, device=dev)
idx = torch.arange(2048, device=dev)
A[idx, idx // 32, idx % 32] = ONE
M = run(A, ones, s127, s127)
rows = M.sum(2).argmax(1); ok_rows = (M.sum(2).max(1).values == 16).all().item()
exp_rows = (torch.arange(2048) // 32) % 16
print("P1 A one-hot: every pattern lights exactly one full row:", ok_rows, "| row == lane%16 :", bool((rows == exp_rows).all()))
# P1b: B one-hot scan, A all ones -> which output column
Bm = torch.zeros(2048, 64, 32, dtype=torch.uint8, device=dev); Bm[idx, idx // 32, idx % 32] = ONE
M = run(ones, Bm, s127, s127)
cols = M.sum(1).argmax(1)
print("P1b B one-hot: col == lane%16 :", bool((cols == exp_rows).all()), "full column:", bool((M.sum(1).max(1).values == 16).all()))
# P2: k mapping: A one-hot at (lane la, byte ja); B one-hot scan -> nonzero iff same k
for la, ja in ((0, 0), (0, 5), (17, 3), (35, 31), (63, 8), (16, 0), (48, 17)):
    A1 = torch.zeros(1, 64, 32, dtype=torch.uint8, device=dev); A1[0, la, ja] = ONE
    M = run(A1, Bm, s127, s127)
    hit = (M.abs().sum((1, 2)) > 0).nonzero().flatten().cpu()
    lanes = sorted(set((hit // 32).tolist())); bytes_ = sorted(set((hit % 32).tolist()))
    print(f"P2 A(lane {la}, byte {ja}) matches B lanes {lanes[:4]}..{lanes[-1]} (n={len(lanes)}) bytes {bytes_}  [hyp: lanes {16*(la//16)}..{16*(la//16)+15}, byte {ja}]")
# P3: scale mapping: all ones, bump lane L's A scale by +1 (x2): which rows change and by how much
sa = torch.full((64, 64), 127, dtype=torch.int32, device=dev); sa[torch.arange(64), torch.arange(64)] = 128
M = run(ones, ones, sa, s127)
base = 128.0
delta = (M[:, :, 0] - base)  # [pattern L][row]
rowsL = delta.argmax(1)
print("P3 scale_a of lane L affects row L%16:", bool((rowsL == torch.arange(64) % 16).all()), "| delta values:", sorted(set(delta.max(1).values.tolist())),
      "| other rows untouched:", bool(((delta != 0).sum(1) == 1).all()))
# which k-block does lane L's scale cover? A zero except k-block q of every row; scale bump lane L -> changes iff q == L//16
for q in range(4):
    Aq = torch.zeros(1, 64, 32, dtype=torch.uint8, device=dev); Aq[0, 16 * q:16 * q + 16, :] = ONE
    M = run(Aq, ones, sa, s127)
    ch = ((M[:, :, 0] - 32.0).abs().sum(1) > 0).nonzero().flatten().tolist()
    print(f"P3b operand k-block held by lanes {16*q}..{16*q+15}: scale bumps that change the result come from lanes {ch[:3]}..{ch[-1] if ch else None} (n={len(ch)})")
# P4: scale in byte 0 only? put 128 in byte 1 with byte0 = 127
sa2 = torch.full((1, 64), 127 | (128 << 8), dtype=torch.int32, device=dev)
M = run(ones, ones, sa2, s127)
print("P4 opsel=0 reads byte 0 only:", bool((M == 128).all()))
print("---- detail: for operand k-block q (held by lanes 16q..16q+15), scale lanes L that change row L%16:")
for q in range(4):
    Aq = torch.zeros(1, 64, 32, dtype=torch.uint8, device=dev); Aq[0, 16 * q:16 * q + 16, :] = ONE
    M = run(Aq, ones, sa, s127)
    d = (M[:, :, 0] - 32.0)
    ch = (d.abs().sum(1) > 0).nonzero().flatten().tolist()
    vals = sorted(set(d[ch].max(1).values.tolist()))
    print(q, ch, vals)
# both bumped
for pair in ((0, 32), (0, 16), (16, 48)):
    sb2 = torch.full((1, 64), 127, dtype=torch.int32, device=dev)
    sa3 = torch.full((1, 64), 127, dtype=torch.int32, device=dev); sa3[0, pair[0]] = 128; sa3[0, pair[1]] = 129
    for q in range(4):
        Aq = torch.zeros(1, 64, 32, dtype=torch.uint8, device=dev); Aq[0, 16 * q:16 * q + 16, :] = ONE
        M = run(Aq, ones, sa3, sb2)
        print("lanes", pair, "scales x2,x4; block", q, "row0 =", M[0, 0, 0].item())
