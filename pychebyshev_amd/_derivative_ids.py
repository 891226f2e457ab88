"""Derivative-spec bookkeeping shared by the interpolant classes: ``get_derivative_id`` and the
"orders xor id" resolution of every evaluation method (reference barycentric.py:1173-1243,
spline.py:447-517, slider.py:201-245 -- three identical copies there)."""
from __future__ import annotations

import numpy as np


class DerivativeIdMixin:
    """Needs ``num_dimensions``, ``max_derivative_order``, ``_derivative_id_registry`` (dict)
    and ``_derivative_id_to_orders`` (list) on the host class."""

    def get_derivative_id(self, derivative_order) -> int:
        """Register a per-dimension derivative-orders tuple; stable, sequential ids from 0."""
        if len(derivative_order) != self.num_dimensions:
            raise ValueError(f"derivative_order length {len(derivative_order)} does not "
                             f"match num_dimensions {self.num_dimensions}")
        for d, o in enumerate(derivative_order):
            if not isinstance(o, (int, np.integer)):
                raise ValueError(f"derivative_order[{d}] must be int, got {type(o).__name__}")
            if o < 0 or o > self.max_derivative_order:
                raise ValueError(f"derivative_order[{d}]={o} out of range [0, {self.max_derivative_order}]")
        key = tuple(int(o) for o in derivative_order)
        found = self._derivative_id_registry.get(key)
        if found is not None:
            return found
        new_id = len(self._derivative_id_to_orders)
        self._derivative_id_registry[key] = new_id
        self._derivative_id_to_orders.append(key)
        return new_id

    def _resolve_derivative_args(self, derivative_order, derivative_id):
        """Exactly one of orders / id: ``ValueError`` for both or neither, ``KeyError`` for an
        unknown id."""
        if derivative_order is not None and derivative_id is not None:
            raise ValueError("provide exactly one of derivative_order or derivative_id, not both")
        if derivative_order is None and derivative_id is None:
            raise ValueError("must provide derivative_order or derivative_id")
        if derivative_id is not None:
            if derivative_id < 0 or derivative_id >= len(self._derivative_id_to_orders):
                raise KeyError(f"unknown derivative_id {derivative_id}; register via get_derivative_id() first")
            return list(self._derivative_id_to_orders[derivative_id])
        return derivative_order
