"""Turn the two counter passes of tools/pmc_sq.sh into the MFMA-utilisation table kept under profiles/.

    python tools/pmc_report.py gpurun_out/pmc_<tag> "<title>" >> profiles/<file>.md

MFMA utilisation = SQ_VALU_MFMA_BUSY_CYCLES / (kernel cycles x 1024 SIMDs), kernel cycles = GRBM_GUI_ACTIVE / 8 (the
counter is summed over the 8 XCDs); SQ_VALU_MFMA_BUSY_CYCLES = (cycles per MFMA) x (MFMA instructions) summed over all
SIMDs (MI355X_MICROARCH.md, PMC units).  Averages are per dispatch over the dispatches of that kernel in the run."""
import collections
import csv
import sys


def load(path):
    agg = collections.defaultdict(lambda: collections.defaultdict(float)); seen = collections.defaultdict(set)
    for r in csv.DictReader(open(path)):
        k = r['Kernel_Name'].split('(')[0]
        agg[k][r['Counter_Name']] += float(r['Counter_Value']); seen[k].add(r['Dispatch_Id'])
    return {k: {c: v / len(seen[k]) for c, v in d.items()} | {'n': len(seen[k])} for k, d in agg.items()}


def main():
    d, title = sys.argv[1], sys.argv[2]
    a, b = load(f'{d}/sq1_counter_collection.csv'), load(f'{d}/sq2_counter_collection.csv')
    print(f'\n## {title}\n')
    print('| kernel | dispatches | kernel cycles | MFMA busy (matrix-pipe utilisation) | VALU / MFMA instructions | LDS instructions | LDS bank-conflict cycles / LDS instruction | wave cycles waiting on LDS |')
    print('|---|---|---|---|---|---|---|---|')
    for k in sorted(a, key=lambda k: -a[k].get('SQ_VALU_MFMA_BUSY_CYCLES', 0)):
        x, y = a[k], b.get(k, {})
        if x.get('SQ_VALU_MFMA_BUSY_CYCLES', 0) <= 0:
            continue
        cyc = x['GRBM_GUI_ACTIVE'] / 8
        print(f"| `{k[:70]}` | {x['n']} | {cyc:.3g} | {100 * x['SQ_VALU_MFMA_BUSY_CYCLES'] / (cyc * 1024):.1f} % | "
              f"{y.get('SQ_INSTS_VALU', 0) / max(y.get('SQ_INSTS_MFMA', 1), 1):.2f} | {y.get('SQ_INSTS_LDS', 0):.3g} | "
              f"{y.get('SQ_LDS_BANK_CONFLICT', 0) / max(y.get('SQ_INSTS_LDS', 1), 1):.2f} | "
              f"{100 * 4 * y.get('SQ_WAIT_INST_LDS', 0) / max(4 * x.get('SQ_WAVE_CYCLES', 1), 1):.1f} % |")


if __name__ == '__main__':
    main()
