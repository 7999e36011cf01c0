"""FASTA / a2m reading and writing plus alignment subsetting: the on-disk formats either side of the Gibbs path
(/root/reference/src/pgen/utils.py:87-168 `parse_fasta`, :221-229 `write_sequential_fasta`, :316-356
`SequenceSubsetter`).  Host-side text handling only; behaviour (cleaning modes, header rule, `>0..n-1` output
names, subsetting semantics) follows the reference and its tests (test/test_utils.py:27-100)."""
import io
import random
import string

_LOWER_DOT_STAR = str.maketrans("", "", string.ascii_lowercase + ".*")
_STAR_DOT_TO_GAP = str.maketrans({"*": None, ".": "-"})
_UNALIGN = str.maketrans("", "", "*.-")


def _records(handle, full_name):
    name, chunks = None, []
    for raw in handle:
        line = raw.strip()
        if not line:
            continue
        if line.startswith(">"):
            if name is not None:
                yield name, "".join(chunks)
            name = line[1:] if full_name else line.split(None, 1)[0][1:]
            chunks = []
        elif name is not None:
            chunks.append(line)
    if name is not None:
        yield name, "".join(chunks)


def parse_fasta(filename, return_names=False, clean=None, full_name=False):
    """Sequences (or (names, sequences)) of a FASTA/a2m file name or open handle.

    clean: None | 'delete' (drop lowercase, '.', '*': a2m insertions) | 'upper' (drop '*', upper-case, '.' -> '-') |
    'unalign' (upper-case, drop '*', '.', '-')."""
    if clean not in (None, "delete", "upper", "unalign"):
        raise ValueError(f"unrecognized input for clean parameter: {clean}")
    own = isinstance(filename, (str, bytes)) or hasattr(filename, "__fspath__")
    handle = open(filename) if own else filename
    try:
        recs = list(_records(handle, full_name))
    finally:
        if own:
            handle.close()
    names = [n for n, _ in recs]
    seqs = [s for _, s in recs]
    if clean == "delete":
        seqs = [s.translate(_LOWER_DOT_STAR) for s in seqs]
    elif clean == "upper":
        seqs = [s.upper().translate(_STAR_DOT_TO_GAP) for s in seqs]
    elif clean == "unalign":
        seqs = [s.upper().translate(_UNALIGN) for s in seqs]
    return (names, seqs) if return_names else seqs


def parse_fasta_string(fasta_string, return_names=False):
    return parse_fasta(io.StringIO(fasta_string), return_names)


def write_sequential_fasta(path, sequences):
    """FASTA whose records are named 0 .. len(sequences)-1 (the reference's `<name>.fasta` output format)."""
    own = isinstance(path, (str, bytes)) or hasattr(path, "__fspath__")
    out = open(path, "w") if own else path
    try:
        for i, seq in enumerate(sequences):
            print(f">{i}\n{seq}", file=out)
    finally:
        if own:
            out.close()


class SequenceSubsetter:
    subset_strategies = {"random", "in_order"}

    @classmethod
    def subset(cls, seq_list, n, keep_first=False, strategy="random", random_seed=None):
        """n members of seq_list: 'in_order' = the first n; 'random' = a shuffle (own Random(random_seed)) then the first n;
        keep_first pins seq_list[0] and samples n-1 from the rest.  n > len returns everything, n <= 0 nothing."""
        if n <= 0:
            return []
        if strategy not in cls.subset_strategies:
            raise ValueError(f"sampler strategy {strategy} not recognized, must be one of {cls.subset_strategies}")
        head, rest = ([seq_list[0]], list(seq_list[1:])) if keep_first else ([], list(seq_list))
        n -= len(head)
        if strategy == "random":
            random.Random(random_seed).shuffle(rest)
        return head + rest[:n]
