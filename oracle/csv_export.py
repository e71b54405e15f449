"""TEST INFRASTRUCTURE — CPU restatement of the reference's `simulation_summary.csv` writer
(utils/csv_export.rs:215-432), for checking eg_export_summary_csv.  Never imported by the product.

Text rules restated from Rust's formatter: `{}` of an f64 prints the shortest digits that round-trip, never an
exponent, and no ".0" on integers; `{:.N}` rounds the exact binary value (Python's format does the same)."""
from decimal import Decimal

from . import api as O

GENERATOR_TYPES = ["OnshoreWind", "OffshoreWind", "DomesticSolar", "CommercialSolar", "UtilitySolar", "Nuclear", "CoalPlant",
                   "GasCombinedCycle", "GasPeaker", "Biomass", "HydroDam", "PumpedStorage", "BatteryStorage", "TidalGenerator",
                   "WaveEnergy"]                                     # models/generator.rs:63-83
OFFSET_TYPES = ["Forest", "Wetland", "ActiveCapture", "CarbonCredit"]    # canonical table order, core.rs:100-114

# yearly row fields (oracle/eg_oracle.h OG_Y_*)
(Y_YEAR, Y_POP, Y_USAGE, Y_GEN, Y_BALANCE, Y_OPINION, Y_YEARLY_CAPITAL, Y_TOTAL_CAPITAL, Y_INFLATION, Y_CO2, Y_OFFSET, Y_NET_CO2,
 Y_YEARLY_CREDIT, Y_TOTAL_CREDIT, Y_YEARLY_SALES, Y_TOTAL_SALES, Y_ACTIVE_GENS, Y_UPGRADE_COSTS, Y_CLOSURE_COSTS,
 Y_YEARLY_TOTAL_COST, Y_TOTAL_COST) = range(21)


def display_f64(v: float) -> str:
    """Rust `format!("{}", v)` for a finite f64."""
    s = format(Decimal(repr(float(v))), "f")
    if "." in s:
        s = s.rstrip("0").rstrip(".")
    return s


def action_row(year: int, action: int) -> str:
    """csv_export.rs:246-393: Year,Action Type,Generator Type,Generator ID,Operation %,Offset Type,Estimated Cost"""
    cost = O.lib().og_action_cost_estimate(action, year)
    kind, gen, op, off = "DoNothing", "", "", ""
    if action < 45:
        kind, gen = "AddGenerator", GENERATOR_TYPES[action // 3]
    elif action < 57:
        kind, off = "AddCarbonOffset", OFFSET_TYPES[(action - 45) // 3]
    elif action == 57:
        kind = "UpgradeEfficiency"
    elif action == 58:
        kind, op = "AdjustOperation", "0"
    elif action == 59:
        kind = "CloseGenerator"
    return f"{year},{kind},{gen},,{op},{off},{cost:.2f}"


def summary_csv_text(metrics, yearly, n_act, act_log, timestamp: str) -> str:
    """The whole file.  metrics[4]; yearly[26][21]; n_act[26]; act_log flat, year-major (SimulationResult.actions)."""
    lines = ["Simulation Summary", f"Timestamp,{timestamp}", "", "Final Metrics",
             f"Final Net Emissions (tonnes CO2),{display_f64(metrics[0])}",
             f"Average Public Opinion (%),{metrics[1] * 100.0:.2f}",
             f"Total Cost (€),{metrics[2]:.2f}",
             f"Power Reliability (%),{metrics[3] * 100.0:.2f}", "",
             "Actions Taken", "Year,Action Type,Generator Type,Generator ID,Operation %,Offset Type,Estimated Cost (€)"]
    pos = 0
    for yi in range(26):
        for _ in range(int(n_act[yi])):
            lines.append(action_row(2025 + yi, int(act_log[pos])))
            pos += 1
    lines += ["", "Yearly Summary Metrics",
              "Year,Population,PowerUsage,PowerGeneration,PowerBalance,PublicOpinion,YearlyCapitalCost,TotalCapitalCost,Inflation,"
              "CO2Emissions,CarbonOffset,NetEmissions,YearlyRevenue,TotalRevenue,ActiveGenerators,YearlyUpgradeCosts,"
              "YearlyClosureCosts,YearlyTotalCost,TotalCost"]
    for yi in range(26):
        r = yearly[yi]
        lines.append(f"{int(r[Y_YEAR])},{int(r[Y_POP])},{r[Y_USAGE]:.2f},{r[Y_GEN]:.2f},{r[Y_BALANCE]:.2f},{r[Y_OPINION]:.4f},"
                     f"{r[Y_YEARLY_CAPITAL]:.2f},{r[Y_TOTAL_CAPITAL]:.2f},{r[Y_INFLATION]:.4f},{r[Y_CO2]:.2f},{r[Y_OFFSET]:.2f},"
                     f"{r[Y_NET_CO2]:.2f},{r[Y_YEARLY_CREDIT]:.2f},{r[Y_TOTAL_CREDIT]:.2f},{int(r[Y_ACTIVE_GENS])},"
                     f"{r[Y_UPGRADE_COSTS]:.2f},{r[Y_CLOSURE_COSTS]:.2f},{r[Y_YEARLY_TOTAL_COST]:.2f},{r[Y_TOTAL_COST]:.2f}")
    return "\n".join(lines) + "\n"
