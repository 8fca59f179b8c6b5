"""Reader of the log that build/performance (and the reference's performance tool) prints: the
fields reference util/plot_performance.py:52-110 takes from it ("Test:", "Block Size:",
"Compression Ratio:", the "Mean:" under "Forward BWT:" / "Inverse BWT:" / "Total Roundtrip:",
"Throughput:", which closes a record) plus this implementation's "GPU ..." lines, which follow
the Throughput line and therefore never reach the reference's parser."""
import re


def _size_kb(text):
    v, unit = text.split()[:2]
    return float(v) * {"B": 1 / 1024.0, "KB": 1.0, "MB": 1024.0, "GB": 1024.0 * 1024.0}[unit]


def _time_ms(text):
    v, unit = text.split()[:2]
    return float(v) * {"s": 1e3, "ms": 1.0, "us": 1e-3, "μs": 1e-3, "µs": 1e-3, "ns": 1e-6}[unit]


def parse_performance_log(text):
    rows, cur, last = [], {}, None
    lines = text.splitlines()
    for i, raw in enumerate(lines):
        line = raw.strip()
        if line.startswith("Test:"):
            cur = {"filename": line.split(":", 1)[1].strip()}
            rows.append(cur)
        elif line.startswith("Block Size:") and cur is not None:
            cur["block_size_kb"] = _size_kb(line.split(":", 1)[1].strip())
        elif line.startswith("Compression Ratio:"):
            cur["compression_ratio"] = float(line.split(":")[1])
        elif line in ("Forward BWT:", "Inverse BWT:", "Total Roundtrip:"):
            key = {"Forward BWT:": "forward_time_ms", "Inverse BWT:": "inverse_time_ms",
                   "Total Roundtrip:": "roundtrip_time_ms"}[line]
            for sub in lines[i + 1:i + 5]:
                if sub.strip().startswith("Mean:"):
                    cur[key] = _time_ms(sub.split("Mean:")[1].split("±")[0].strip())
                    break
        elif line.startswith("Throughput:"):
            cur["throughput_mb_s"] = float(line.split(":")[1].split()[0])
        elif line.startswith("GPU Roundtrip Rate:"):
            cur["gpu_roundtrip_gbps"] = float(line.split(":")[1].split()[0])
        elif line.startswith("GPU HBM Roofline:"):
            cur["gpu_pct_hbm"] = float(line.split(":")[1].split()[0])
        elif line.startswith("GPUs Used:"):
            cur["gpus_used"] = int(re.findall(r"\d+", line)[0])
    return [r for r in rows if "throughput_mb_s" in r]
