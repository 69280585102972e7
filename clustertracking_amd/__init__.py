"""clustertracking_amd -- MI355X-native engine for the per-cluster least-squares
refinement of caspervdw/clustertracking (``refine_leastsq``).

Public names follow the reference package (``clustertracking/__init__.py:10-18``)
for the part of it that this engine accelerates.
"""
import logging

from .refine import refine_leastsq, prepare_batch, write_back
from .find import find_clusters
from .fitfunc import FitFunctions
from .utils import ArrayReader, RefineException
from . import constraints, artificial, link

link_df = link.link

__all__ = ['refine_leastsq', 'find_clusters', 'link', 'link_df', 'FitFunctions', 'constraints',
           'artificial', 'ArrayReader', 'RefineException', 'prepare_batch',
           'write_back']

logger = logging.getLogger(__name__)
logger.addHandler(logging.NullHandler())
