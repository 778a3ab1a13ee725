#!/usr/bin/env python3
"""Derives the SIRV transcript tables (BASELINE config[0]) from the reference's test_data GTFs.

The reference ships no BAM and no reads (test_data/ holds the SIRV FASTA, three GTF annotations
C/I/O = 69/44/100 transcripts, and two truth spreadsheets), so config[0] "SIRV test_data BAM" is
exercised on a SIRV-shaped synthetic store: per annotation, transcript -> (gene, length = sum of
exon lengths, exon-sharing neighbours).  This script extracts that DATA (ids, genes, lengths,
pairwise exonic overlap in bases) into tests/golden/sirv_txps.json; nothing of the GTF text is kept.
Run here (needs /root/reference):  python scripts/make_sirv_fixture.py
"""
import json, os, re, collections
REF = "/root/reference/test_data"
OUT = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden", "sirv_txps.json")
out = {}
for tag in "CIO":
    path = os.path.join(REF, f"SIRV_isoforms_multi-fasta-annotation_{tag}_170612a.gtf")
    exons = collections.OrderedDict()
    gene = {}
    for line in open(path):
        f = line.rstrip("\n").split("\t")
        if len(f) < 9 or f[2] != "exon":
            continue
        g = re.search(r'gene_id "([^"]+)"', f[8]).group(1)
        t = re.search(r'transcript_id "([^"]+)"', f[8]).group(1)
        exons.setdefault(t, []).append((int(f[3]), int(f[4])))
        gene[t] = g
    names = list(exons)
    genes = sorted(set(gene.values()))
    def overlap(a, b):
        tot = 0
        for s1, e1 in exons[a]:
            for s2, e2 in exons[b]:
                tot += max(0, min(e1, e2) - max(s1, s2) + 1)
        return tot
    ov = {}
    for i, a in enumerate(names):
        for j, b in enumerate(names):
            if i < j and gene[a] == gene[b]:
                o = overlap(a, b)
                if o:
                    ov[f"{i},{j}"] = o
    out[tag] = dict(names=names, gene=[genes.index(gene[t]) for t in names],
                    length=[sum(e - s + 1 for s, e in exons[t]) for t in names], overlap=ov)
    print(tag, len(names), "transcripts", len(genes), "genes", len(ov), "overlapping pairs")
json.dump(out, open(OUT, "w"))
print("wrote", OUT, os.path.getsize(OUT), "bytes")
