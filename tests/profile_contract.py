"""What `kmcp profile` (and `kmcp utils filter` / `merge-regions`, which share the parser) reads from a `kmcp search` result —
the consumer side of the TSV contract, restated for the tests (test infrastructure, not product):

  * kmcp/cmd/profile.go:1939-1962 (same loop at :739-772): blank lines are ignored; a line starting with '#' is matched against
    `^# ([\\w ]+): (.+)` (merge.go:460) and, if it is "input queries", its integer value is added to the total; every other
    '#' line (the header row) is dropped; everything else is a match row;
  * kmcp/cmd/util-profile.go:94-182 parseMatchResult with numFields = 13 (profile.go:728): the row is cut at the first 12 tabs
    (util.go:257-277 stringSplitNByByte), fewer than 13 pieces is fatal; qCov = ParseFloat(items[11]) — rows below
    --min-query-cov are dropped before anything else is parsed; FPR = ParseFloat(items[3]) — rows above --max-fpr are dropped;
    then Query = items[0], qLen / qKmers / hits / chunkIdx / chunks / kSize / mKmers = strconv.Atoi (optional sign, decimal
    digits only, nothing else), tLen = ParseUint(…, 10, 64), Target = items[5].

A result that this reader accepts, with the values the oracle computed, drops in under `kmcp profile`."""
import re

RE_STATS = re.compile(r"^# ([\w ]+): (.+)")
_ATOI = re.compile(r"^[+-]?[0-9]+$")
_UINT = re.compile(r"^[0-9]+$")
# strconv.ParseFloat: decimal / exponent forms, inf, nan (hex floats and underscores are not produced by anybody here)
_FLOAT = re.compile(r"^[+-]?(?:(?:[0-9]+\.?[0-9]*|\.[0-9]+)(?:[eE][+-]?[0-9]+)?|inf|infinity|nan)$", re.I)


class ContractError(ValueError):
    pass


def _atoi(s, what):
    if not _ATOI.match(s):
        raise ContractError(f"failed to parse {what}: {s}")
    return int(s)


def _float(s, what):
    if not _FLOAT.match(s):
        raise ContractError(f"failed to parse {what}: {s}")
    return float(s)


def split_n(line, sep="\t", n=13):
    """util.go:257-277: at most n pieces, the last one keeps the rest of the line."""
    return line.split(sep, n - 1)


def parse_match_result(line, max_fpr=0.05, min_qcov=0.55, num_fields=13):
    items = split_n(line, "\t", num_fields)
    if len(items) < num_fields:
        raise ContractError("invalid kmcp search result format")
    qcov = _float(items[11], "qCov")
    if qcov < min_qcov:
        return None
    fpr = _float(items[3], "FPR")
    if fpr > max_fpr:
        return None
    if not _UINT.match(items[8]) or int(items[8]) >= 1 << 64:
        raise ContractError(f"failed to parse genomeSize: {items[8]}")
    return {"query": items[0], "qlen": _atoi(items[1], "qLen"), "qkmers": _atoi(items[2], "qKmers"), "fpr": fpr, "hits": _atoi(items[4], "hits"),
            "target": items[5], "chunk_idx": _atoi(items[6], "chunkIdx"), "chunks": _atoi(items[7], "IdxNum"), "gsize": int(items[8]),
            "k": _atoi(items[9], "K"), "mkmers": _atoi(items[10], "mKmers"), "qcov": qcov}


def read_search_result(text, max_fpr=0.05, min_qcov=0.55):
    """-> (list of match dicts that pass the two filters, total of the `# input queries:` lines, all stats as a dict)"""
    matches, total, stats = [], 0, {}
    for line in text.split("\n"):
        line = line.rstrip("\r")
        if line == "":
            continue
        if line[0] == "#":
            m = RE_STATS.match(line)
            if m:
                stats[m.group(1)] = m.group(2)
                if m.group(1) == "input queries":
                    total += _atoi(m.group(2), "input queries")
            continue
        r = parse_match_result(line, max_fpr, min_qcov)
        if r is not None:
            matches.append(r)
    return matches, total, stats
