"""TEST INFRASTRUCTURE ONLY - minimal data-holder stand-in for ``ase`` (not installed
here) so that the true reference's ``SiteNetwork`` can be constructed by
``oracle/make_fixtures.py``.  Holds positions / numbers / cell; performs no
arithmetic that the landmark path depends on."""
import numpy as np


class Atoms(object):
    def __init__(self, positions=None, numbers=None, cell=None, symbols=None, pbc=True):
        self.positions = np.array(positions, dtype=np.float64).reshape(-1, 3)
        n = len(self.positions)
        self.numbers = np.zeros(n, dtype=np.int64) if numbers is None else np.array(numbers, dtype=np.int64)
        self.cell = np.zeros((3, 3)) if cell is None else np.array(cell, dtype=np.float64)
        self.pbc = pbc

    def __len__(self):
        return len(self.positions)

    def copy(self):
        return Atoms(self.positions.copy(), self.numbers.copy(), self.cell.copy(), pbc=self.pbc)

    def __delitem__(self, key):
        key = np.asarray(key)
        if key.dtype == bool:
            keep = ~key
        else:
            keep = np.ones(len(self), dtype=bool)
            keep[key] = False
        self.positions = self.positions[keep]
        self.numbers = self.numbers[keep]

    def __getitem__(self, key):
        return Atoms(self.positions[key], self.numbers[key], self.cell.copy(), pbc=self.pbc)

    def get_positions(self):
        return self.positions.copy()

    def get_atomic_numbers(self):
        return self.numbers.copy()

    def get_masses(self):
        return np.ones(len(self))

    def get_cell(self):
        return self.cell.copy()

    def extend(self, other):
        self.positions = np.concatenate([self.positions, other.positions])
        self.numbers = np.concatenate([self.numbers, other.numbers])
