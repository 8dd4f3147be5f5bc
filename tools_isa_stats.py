#!/usr/bin/env python3
"""Per-kernel ISA hygiene table from a `hipcc -S --cuda-device-only` listing: registers, spills,
scratch, addressing form of the wide loads (64-bit VGPR pair vs saddr + 32-bit voffset), 64-bit / quarter-rate integer
VALU, packed-f32 VALU (v_pk_{add,mul,fma}_f32) between the first and the last MFMA of the kernel, SGPR-spill lane ops.

usage: tools_isa_stats.py <listing.s> [...]"""
import re
import sys

P_V64 = r'global_load_dwordx4 v\[\d+:\d+\], v\[\d+:\d+\], off'
P_SADDR = r'global_load_dwordx4 v\[\d+:\d+\], v\d+, s\['
P_PK = r'v_pk_(add|mul|fma)_f32'


def main():
    for path in sys.argv[1:]:
        txt = open(path).read()
        parts = re.split(r'\n(_Z\w+):[^\n]*\n', txt)
        meta = {}
        for blk in txt.split('  - .agpr_count:')[1:]:
            name = re.search(r'\.name:\s+(\S+)', blk).group(1)
            g = lambda k: re.search(r'\.%s:\s+(\d+)' % k, blk).group(1)   # noqa: E731
            meta[name] = dict(sc=g('private_segment_fixed_size'), vg=g('vgpr_count'), ss=g('sgpr_spill_count'), vs=g('vgpr_spill_count'))
        for i in range(1, len(parts), 2):
            name, body = parts[i], parts[i + 1]
            if name not in meta:
                continue
            lines = body.split('.Lfunc_end')[0].split('\n')
            mf = [k for k, l in enumerate(lines) if 'v_mfma' in l]
            inner = lines[mf[0]:mf[-1] + 1] if mf else []
            c = lambda pat, ls=lines: sum(1 for l in ls if re.search(pat, l))   # noqa: E731
            short = re.sub(r'Ev11DffModelDev10DffRunArgs', '', name)[3:70]
            m = meta[name]
            print("%-62s vgpr=%3s vgpr_spill=%3s sgpr_spill=%3s scratch=%4sB | ldx4 vaddr64=%3d saddr=%3d | mad_u64=%3d lshl_add_u64=%3d mul_lo=%3d | "
                  "pk_f32 between MFMAs=%3d (all %3d) | lane-ops=%3d mfma=%d" % (
                      short, m['vg'], m['vs'], m['ss'], m['sc'], c(P_V64), c(P_SADDR), c('v_mad_u64_u32'), c('v_lshl_add_u64'),
                      c('v_mul_lo_u32'), c(P_PK, inner), c(P_PK), c('v_(read|write)lane'), len(mf)))


if __name__ == "__main__":
    main()
