"""Per-wave instruction mix of the conv kernels from a tools/pmc_any.sh summary (gpurun_out/pmcany_<tag>.txt)."""
import re
import sys

t = open(sys.argv[1]).read()
for b in re.split(r'\n(?=\S)', t):
    lines = b.strip().split('\n')
    name = lines[0]
    if 'gather_h' not in name and 'wgrad_h' not in name:
        continue
    d = {}
    for l in lines[1:]:
        q = l.split()
        d[q[0]] = float(q[1])
    m = re.search(r'(gather_h_kernel|wgrad_h_kernel)<([^>]*)>.*grid=(\d+)', name)
    waves = int(m.group(3)) / 64
    g = lambda k: d.get(k, 0.0)
    print('%s<%s> grid=%s | per wave: MFMA %.0f VALU %.0f SALU %.0f LDS %.0f VMEM %.0f SMEM %.0f | wave kcyc %.1f wait_any %.0f%% wait_inst %.0f%% '
          'active %.0f%% | GRBM %.0fk mfma busy %.2f' % (
              m.group(1)[:8], m.group(2), m.group(3), g('SQ_INSTS_MFMA') / waves, g('SQ_INSTS_VALU') / waves, g('SQ_INSTS_SALU') / waves,
              g('SQ_INSTS_LDS') / waves, g('SQ_INSTS_VMEM_RD') / waves, g('SQ_INSTS_SMEM') / waves, g('SQ_WAVE_CYCLES') * 4 / waves / 1e3,
              100 * g('SQ_WAIT_ANY') / max(g('SQ_WAVE_CYCLES'), 1), 100 * g('SQ_WAIT_INST_ANY') / max(g('SQ_WAVE_CYCLES'), 1),
              100 * g('SQ_ACTIVE_INST_ANY') / max(g('SQ_WAVE_CYCLES'), 1), g('GRBM_GUI_ACTIVE') / 1e3,
              g('SQ_VALU_MFMA_BUSY_CYCLES') / max(g('GRBM_GUI_ACTIVE') / 8 * 256 * 4, 1)))
