# LDS cycle model of ds_read_b128 on gfx950 (MI355X_MICROARCH.md, LDS): four lane groups, 64 banks of 4 B, N-way = N cycles per group
G128 = [list(range(0,4))+list(range(12,16))+list(range(20,28)), list(range(4,12))+list(range(16,20))+list(range(28,32)),
        list(range(32,36))+list(range(44,48))+list(range(52,60)), list(range(36,44))+list(range(48,52))+list(range(60,64))]
def cycles_b128(addr):      # addr[lane] in floats (16-B aligned)
    tot = 0
    for g in G128:
        banks = {}
        for l in g:
            for w in range(4):
                b = (addr[l] + w) % 64
                banks.setdefault(b, set()).add(addr[l] + w)
        tot += max(len(v) for v in banks.values())
    return tot
def lanes():
    return [(l & 15, l >> 4) for l in range(64)]
# 1. stream / sq kernels: weights [row][SKP=132], lane (j,q) reads row 16a+j, k = 16kc+4q
for P in (132, 136, 144):
    print("sq weights pitch", P, cycles_b128([j*P + 4*q for j, q in lanes()]))
# k-quad-major image [K/4][N][4]
print("sq weights [k/4][n][4]", cycles_b128([(q)*4*128 + j*4 for j, q in lanes()]))
# S0P = 20: sW0 + (16a+j)*S0P + 4q
print("sW0 pitch 20", cycles_b128([j*20 + 4*q for j, q in lanes()]))
# 2. tiled kernel, plain [row][PK=36] form: row*PK + 16kc + 4q
print("tiled PK=36", cycles_b128([j*36 + 4*q for j, q in lanes()]))
def tr_quad(row, c4): return 4 * (c4 ^ ((row >> 3) & 7))
for base in (0, 16, 32, 48):
    for kc in (0, 1):
        print("tiled TR base", base, "kc", kc, cycles_b128([(base+j)*36 + tr_quad(base+j, 4*kc+q) for j, q in lanes()]))
print("---- new layouts")
def sig(row, c4): return c4 ^ (((row >> 1) ^ (row >> 4)) & 7)
for base in (0, 16, 32, 48, 64, 112):
    for kc in (0, 1):
        print("tiled new base", base, "kc", kc, cycles_b128([(base+j)*32 + 4*sig(base+j, 4*kc+q) for j, q in lanes()]))
# writes: ds_write_b128, 8 groups of 8 contiguous lanes, banks mod 32
def cycles_w128(addr):
    tot = 0
    for g in range(8):
        banks = {}
        for l in range(8*g, 8*g+8):
            for w in range(4):
                banks.setdefault((addr[l]+w) % 32, set()).add(addr[l]+w)
        tot += max(len(v) for v in banks.values())
    return tot
for wave in range(4):
    for c in range(4):
        tids = [64*wave + l for l in range(64)]
        a_new = [ (4*(t%32)+c)*32 + 4*sig(4*(t%32)+c, t//32) for t in tids]
        a_old = [ (4*(t%32)+c)*36 + 4*((t//32) ^ (((4*(t%32)+c)>>3)&7)) for t in tids]
        print("store_tile_tr wave", wave, "c", c, "new", cycles_w128(a_new), "old", cycles_w128(a_old))
# plain store_tile<ROWS,false>: idx -> row = idx/8, quad = idx%8
tids = list(range(64))
print("store_tile plain new", cycles_w128([(t//8)*32 + 4*sig(t//8, t%8) for t in tids]), "old", cycles_w128([(t//8)*36 + 4*(t%8) for t in tids]))
# sq image reads with the XOR layout
print("sq image xor", [cycles_b128([(16*a+j)*128 + 4*((4*kc+q) ^ j) for j, q in lanes()]) for a in (0,3) for kc in (0,1,5)])
# sW0 pitch 16 with quad ^ (((row>>3)&1)<<1)
print("sW0 new", cycles_b128([j*16 + 4*(q ^ (((j>>3)&1)<<1)) for j, q in lanes()]))
