"""Minimal Newick reader/writer for the Python host (tests, bench).

Dialect: nested parentheses, optional labels (plain or 'single quoted'), optional
':branch length', '[comments]' skipped, trees separated by ';'. Replaces the slice of
genesis (DefaultTreeNewickReader / NewickInputIterator) the reference uses for input
(QuartetScores.cpp:97-98, QuartetCounterLookup.hpp:202-206). The production host is the
C++ one under quartetscores_amd/csrc/host; this module mirrors it for the harness.
"""
from __future__ import annotations

from dataclasses import dataclass, field
from typing import Iterator, List, Optional


@dataclass
class Node:
    name: str = ""
    length: Optional[str] = None  # kept verbatim for the writer
    children: List["Node"] = field(default_factory=list)
    parent: Optional["Node"] = None
    index: int = -1

    @property
    def is_leaf(self):
        return not self.children


class NewickError(ValueError):
    pass


def _skip(s, i):
    n = len(s)
    while i < n:
        ch = s[i]
        if ch in " \t\r\n":
            i += 1
        elif ch == "[":
            j = s.find("]", i)
            if j < 0:
                raise NewickError("unterminated comment")
            i = j + 1
        else:
            break
    return i


def _label(s, i):
    i = _skip(s, i)
    n = len(s)
    if i < n and s[i] == "'":
        i += 1
        out = []
        while i < n:
            if s[i] == "'":
                if i + 1 < n and s[i + 1] == "'":
                    out.append("'")
                    i += 2
                    continue
                i += 1
                break
            out.append(s[i])
            i += 1
        return "".join(out), i
    j = i
    while j < n and s[j] not in "(),:;[ \t\r\n":
        j += 1
    return s[i:j], j


def _parse_one(s, i):
    """Iterative parser (deep caterpillar trees must not hit the recursion limit)."""
    root = Node()
    cur = root
    i = _skip(s, i)
    n = len(s)
    if i < n and s[i] != "(":
        # single-leaf tree
        root.name, i = _label(s, i)
    while i < n:
        i = _skip(s, i)
        if i >= n:
            break
        ch = s[i]
        if ch == "(":
            child = Node(parent=cur)
            cur.children.append(child)
            cur = child
            i += 1
            i = _skip(s, i)
            if i < n and s[i] not in "(,)":
                cur.name, i = _label(s, i)
        elif ch == ",":
            if cur.parent is None:
                raise NewickError("',' outside parentheses")
            sib = Node(parent=cur.parent)
            cur.parent.children.append(sib)
            cur = sib
            i += 1
            i = _skip(s, i)
            if i < n and s[i] not in "(,)":
                cur.name, i = _label(s, i)
        elif ch == ")":
            if cur.parent is None:
                raise NewickError("unbalanced ')'")
            cur = cur.parent
            i += 1
            i = _skip(s, i)
            if i < n and s[i] not in "(),:;":
                cur.name, i = _label(s, i)
        elif ch == ":":
            i += 1
            i = _skip(s, i)
            j = i
            while j < n and s[j] not in "(),;[ \t\r\n":
                j += 1
            cur.length = s[i:j]
            i = j
        elif ch == ";":
            i += 1
            break
        else:
            raise NewickError(f"unexpected character {ch!r} at offset {i}")
    if cur is not root:
        raise NewickError("unbalanced '('")
    return root, i


def parse_trees(text: str) -> Iterator[Node]:
    i = 0
    n = len(text)
    while True:
        i = _skip(text, i)
        if i >= n:
            return
        root, i = _parse_one(text, i)
        yield root


def parse_tree(text: str) -> Node:
    for t in parse_trees(text):
        return t
    raise NewickError("empty input")


def preorder(root: Node) -> List[Node]:
    out, stack = [], [root]
    while stack:
        x = stack.pop()
        out.append(x)
        stack.extend(reversed(x.children))
    return out


def _quote(name: str) -> str:
    if name and any(ch in name for ch in " \t\r\n()[]':;,"):
        return "'" + name.replace("'", "''") + "'"
    return name


def write(root: Node, comment=None) -> str:
    """Newick text; comment(node) -> str|None is appended as [..] after the branch length.

    Layout of an element: name, ':length' (only when the input had one), '[comment]'.
    The reference's output layout is decided by genesis v0.16.0's NewickWriter, which is
    absent here (parity unpinned); this is our own pinned format.
    """
    parts = []
    stack = [(root, 0)]
    while stack:
        node, state = stack.pop()
        if state == 0:
            if node.children:
                parts.append("(")
                stack.append((node, 1))
                for k, ch in reversed(list(enumerate(node.children))):
                    stack.append((ch, 0))
                    if k > 0:
                        stack.append((None, 2))
                continue
            state = 1
        if state == 2:
            parts.append(",")
            continue
        if node.children:
            parts.append(")")
        parts.append(_quote(node.name))
        if node.length is not None:
            parts.append(":" + node.length)
        if comment is not None:
            cm = comment(node)
            if cm:
                parts.append("[" + cm + "]")
    return "".join(parts) + ";"
