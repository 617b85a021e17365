#!/usr/bin/env python3
"""Digest of the product's sources (clustering_amd/csrc + include/), comments and blank lines left out.

One definition for everyone who ties a number to a build:
  * the Makefile embeds it in the library (`dc_hip_build_digest()`, via lib/obj/dc_build_digest.h),
  * bench.py prints the LIBRARY's digest in its line and refuses counter profiles measured on another one,
  * scratch/make_pmc_profile.py records it in profiles/*_pmc.json.
A comment edit does not change it, so it does not force the profiles to be regenerated.

usage: digest.py            print the digest
       digest.py HEADER     write `#define DC_BUILD_DIGEST "<digest>"` to HEADER if its content would change"""
import hashlib
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
DIRS = (HERE, os.path.normpath(os.path.join(HERE, "..", "..", "include")))
EXTS = (".hip", ".hpp", ".cpp", ".h")


def strip_comments(text):
    """C / C++ source without // and /* */ comments (string and character literals respected), without trailing blanks
    and without empty lines."""
    out = []
    i, n = 0, len(text)
    while i < n:
        ch = text[i]
        if ch == '"' or ch == "'":
            j = i + 1
            while j < n and text[j] != ch:
                j += 2 if text[j] == "\\" else 1
            out.append(text[i:j + 1])
            i = j + 1
        elif text.startswith("//", i):
            j = text.find("\n", i)
            i = n if j < 0 else j
        elif text.startswith("/*", i):
            j = text.find("*/", i + 2)
            out.append(" ")
            i = n if j < 0 else j + 2
        else:
            out.append(ch)
            i += 1
    lines = (l.rstrip() for l in "".join(out).split("\n"))
    return "\n".join(l for l in lines if l)


def strip_make_comments(text):
    lines = (l.split("#", 1)[0].rstrip() if not l.startswith("\t") else l.rstrip() for l in text.split("\n"))
    return "\n".join(l for l in lines if l)


def source_digest(dirs=DIRS):
    h = hashlib.sha256()
    for base in dirs:
        for f in sorted(os.listdir(base)):
            path = os.path.join(base, f)
            if f.endswith(EXTS):
                body = strip_comments(open(path, encoding="utf-8", errors="replace").read())
            elif f == "Makefile":
                body = strip_make_comments(open(path, encoding="utf-8", errors="replace").read())
            else:
                continue
            h.update(f.encode())
            h.update(b"\0")
            h.update(body.encode())
            h.update(b"\0")
    return h.hexdigest()[:16]


if __name__ == "__main__":
    d = source_digest()
    if len(sys.argv) > 1:
        text = '#define DC_BUILD_DIGEST "%s"\n' % d
        try:
            same = open(sys.argv[1]).read() == text
        except OSError:
            same = False
        if not same:
            os.makedirs(os.path.dirname(os.path.abspath(sys.argv[1])), exist_ok=True)
            open(sys.argv[1], "w").write(text)
    else:
        print(d)
