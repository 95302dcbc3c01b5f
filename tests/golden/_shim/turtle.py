"""transforms/misc.py:1 does `from turtle import forward`, which needs tkinter
(absent).  The name is never used."""


def forward(*a, **k):
    raise NotImplementedError
