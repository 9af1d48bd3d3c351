# register / spill / LDS / scratch figures of the device kernels in an object file: tools/kernel_regs.sh molar_amd/csrc/pair_k0.o [name filter]
T=$(mktemp -d)
/opt/rocm/lib/llvm/bin/llvm-objcopy -O binary --only-section=.hip_fatbin $1 $T/fat.bin
/opt/rocm/lib/llvm/bin/clang-offload-bundler --unbundle --type=o --input=$T/fat.bin --targets=hipv4-amdgcn-amd-amdhsa--gfx950 --output=$T/dev.o
/opt/rocm/lib/llvm/bin/llvm-readelf --notes $T/dev.o | python3 -c "
import sys,re
txt=sys.stdin.read()
flt=sys.argv[1] if len(sys.argv)>1 else ''
for blk in txt.split('- .agpr_count')[1:]:
    g=lambda k: (re.search(r'\.'+k+r':\s+(\S+)', blk) or [None,'?'])[1]
    name=g('name')
    if flt in name:
        print(name[:100], 'vgpr',g('vgpr_count'),'spill',g('vgpr_spill_count'),'sgpr',g('sgpr_count'),'sspill',g('sgpr_spill_count'),'lds',g('group_segment_fixed_size'),'scratch',g('private_segment_fixed_size'))
" "$2"
rm -rf $T
