"""geneticscre_amd -- MI355X-native permutation-tested path-join scorer (drop-in for geneticsCRE's JoinExec path)."""
__version__ = "0.1.0"
