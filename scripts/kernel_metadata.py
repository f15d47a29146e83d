"""Register and spill figures of every rtow_trace_kernel instance, read from the code-object metadata in the ISA
listing that `make -C raytracinginoneweekend_amd/csrc asm` writes.  usage: kernel_metadata.py [listing.s]"""
import os, re, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
path = sys.argv[1] if len(sys.argv) > 1 else os.path.join(ROOT, "raytracinginoneweekend_amd", "lib", "asm", "rtow_kernels.s")
text = open(path).read()
meta = text[text.index("amdhsa.kernels:"):]
print(f"{'kernel instance <K_LDS, X_LDS, SLAB_AXIS>':44s} vgpr  vgpr_spill  sgpr  sgpr_spill  scratch_B  lds_static")
for block in meta.split("  - .agpr_count")[1:]:
    get = lambda k: re.search(r"\.%s:\s+(\S+)" % k, block).group(1)
    name = get("name")
    m = re.match(r"_ZN4rtow17rtow_trace_kernelILb(\d)ELb(\d)ELi(n?\d)EEE", name)
    if not m:
        continue
    inst = f"rtow_trace_kernel<{m.group(1)}, {m.group(2)}, {m.group(3).replace('n', '-')}>"
    print(f"{inst:44s} {get('vgpr_count'):>4s}  {get('vgpr_spill_count'):>10s}  {get('sgpr_count'):>4s}  {get('sgpr_spill_count'):>10s}  "
          f"{get('private_segment_fixed_size'):>9s}  {get('group_segment_fixed_size'):>10s}")
