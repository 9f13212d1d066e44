"""Stand-in for the PyPI package `trampoline` (absent from this image; the reference needs exactly two
symbols of it, torchsde/_brownian/brownian_interval.py:16,183,275-315). Used ONLY by make_golden.py to import
the reference in the dev container. Semantics: run a generator-based recursion on an explicit stack;
`yield gen` calls a sub-generator and sends its return value back; `raise TailCall(gen)` replaces the
current frame."""


class TailCall(Exception):
    def __init__(self, gen):
        super().__init__()
        self.gen = gen


def trampoline(gen):
    stack = [gen]
    value = None
    while stack:
        top = stack[-1]
        try:
            child = top.send(value)
            value = None
            stack.append(child)
        except StopIteration as stop:
            stack.pop()
            value = stop.value
        except TailCall as tc:
            stack.pop()
            stack.append(tc.gen)
            value = None
    return value
