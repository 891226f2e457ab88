"""Small accessor surface shared by the four interpolant classes (the reference repeats
these per class: barycentric.py:1118-1431, tensor_train.py:2721-2806, spline.py:904-992,
slider.py:428-505)."""
from __future__ import annotations

import copy


class ErgonomicsMixin:
    """descriptor / introspection getters and ``clone``; classes add what depends on their grids."""

    def get_constructor_type(self) -> str:
        return type(self).__name__

    def set_descriptor(self, descriptor: str) -> None:
        if not isinstance(descriptor, str):
            raise TypeError(f"descriptor must be str, got {type(descriptor).__name__}")
        self.descriptor = descriptor

    def get_descriptor(self) -> str:
        return self.descriptor

    def get_max_derivative_order(self) -> int:
        return self.max_derivative_order

    @staticmethod
    def is_dimensionality_allowed(num_dimensions: int) -> bool:
        return isinstance(num_dimensions, int) and num_dimensions >= 1

    def clone(self):
        """Independent deep copy.  Like save/load it goes through the pickle state: the source
        callable and any device handle are not carried over (``function`` is ``None``; the copy
        uploads its own model on first evaluation)."""
        return copy.deepcopy(self)
