"""rrmpg_amd.models -- same model classes as rrmpg.models of the reference
(reference: rrmpg/models/__init__.py:11-18), backed by the gfx950 kernels."""

from .abcmodel import ABCModel
from .hbvedu import HBVEdu
from .gr4j import GR4J
from .cemaneige import Cemaneige
from .cemaneigegr4j import CemaneigeGR4J
from .cemaneigehystgr4j import CemaneigeHystGR4J
from .cemaneigegr4jice import CemaneigeGR4JIce
from .cemaneigehystgr4jice import CemaneigeHystGR4JIce

__all__ = ["ABCModel", "HBVEdu", "GR4J", "Cemaneige", "CemaneigeGR4J",
           "CemaneigeHystGR4J", "CemaneigeGR4JIce", "CemaneigeHystGR4JIce"]
