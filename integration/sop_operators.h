/* sop_operators.h -- the operator table a drop-in build must present to Houdini: type name, label, input counts and
 * parameter interface of the reference's five SOPs, restated as data (plain C, no HDK) with the reference lines they come
 * from. The SOP classes themselves (OP_Operator registration, SOP_NodeVerb subclasses, PRM_TemplateBuilder over the .ds
 * text, GU_PrimVDB I/O) stay the reference's own files: they contain no GPU code and need only the edits listed in
 * hns_shim.cpp. COMPILE NOTE: this header compiles anywhere; tests/test_integration.py checks it against the counts below.
 * Paths are relative to the reference checkout. */
#ifndef HNS_SOP_OPERATORS_H
#define HNS_SOP_OPERATORS_H

typedef struct {
	const char* name;     /* parm token */
	const char* label;
	const char* type;     /* .ds type: float | integer | string | toggle */
	const char* def;      /* default expression, "" = none given */
	const char* range;    /* .ds range, "" = none */
} hns_sop_parm;

typedef struct {
	const char* type_name; /* OP_Operator internal name */
	const char* label;     /* UI label = SOP_NodeVerb::name() */
	int min_inputs, max_inputs;
	const char* flags;     /* OP_FLAG_* */
	const char* entry_points; /* the extern "C" functions its cook calls */
	int n_parms;
	const hns_sop_parm* parms;
} hns_sop_operator;

/* src/SOP/HNanoSolver/SOP_HNanoSolver.cpp:21-88 */
static const hns_sop_parm hns_parms_hnanosolver[] = {
    {"timestep", "Time Step", "float", "1/$FPS", ""},
    {"padding", "Voxel Padding", "integer", "", "1! 100"},
    {"iterations", "Pressure Projection", "integer", "", "1! 100"},
    {"expansion_rate", "Expansion Rate", "float", "0.1", ""},
    {"temperature_gain", "Temperature Gain", "float", "0.5", ""},
    {"buoyancy_strength", "Buoyancy Strength", "float", "1.0", ""},
    {"ambient_temp", "Ambient Temperature", "float", "23.0", ""},
    {"vorticity", "Vorticity Scale", "float", "1", ""},
    {"factor_scale", "Vorticity Factor Scale", "float", "0.5", ""},
};
/* src/SOP/Advection/SOP_VDBAdvect.cpp:21-49 */
static const hns_sop_parm hns_parms_hnanoadvect[] = {
    {"agroup", "Density Volumes", "string", "", ""},
    {"bgroup", "Velocity Volume", "string", "", ""},
    {"timestep", "Time Step", "float", "1/$FPS", ""},
};
/* src/SOP/VelocityAdvection/SOP_VDBAdvectVelocity.cpp:24-44 */
static const hns_sop_parm hns_parms_hnanoadvectvelocity[] = {
    {"agroup", "Velocity Volumes Advected", "string", "", ""},
    {"timestep", "Time Step", "float", "1/$FPS", ""},
};
/* src/SOP/ProjectNonDivergent/SOP_VDBProjectNonDivergent.cpp:18-58 */
static const hns_sop_parm hns_parms_hnanoprojectnondivergent[] = {
    {"velgrid", "Velocity Volumes", "string", "", ""},
    {"voxelsize", "Voxel Size", "float", "0.5", ""},
    {"iterations", "Iterations", "integer", "", "1! 100"},
    {"outdiv", "Output Divergence", "toggle", "0", ""},
};
/* src/SOP/ReadWrite/SOP_VDBFromGrid.cpp:20-58 (no GPU entry point: VDB <-> flat round trip through IndexGridBuilder) */
static const hns_sop_parm hns_parms_hnanofromgrid[] = {
    {"agroup", "Density Volumes", "string", "", ""},
    {"bgroup", "Velocity Volumes", "string", "", ""},
    {"timestep", "Time Step", "float", "1/$FPS", ""},
};

static const hns_sop_operator hns_sop_operators[] = {
    /* src/SOP/HNanoSolver/SOP_HNanoSolver.cpp:15-18; inputs: feedback VDBs, source VDBs, [collision SDF] (:106-108) */
    {"hnanosolver", "HNanoSolver", 2, 3, "OP_FLAG_GENERATOR", "CreateIndexGrid, Compute_Sim", 9, hns_parms_hnanosolver},
    /* src/SOP/Advection/SOP_VDBAdvect.cpp:15-18 */
    {"hnanoadvect", "HNanoAdvect", 2, 2, "OP_FLAG_GENERATOR", "AdvectIndexGrid", 3, hns_parms_hnanoadvect},
    /* src/SOP/VelocityAdvection/SOP_VDBAdvectVelocity.cpp:18-21 */
    {"hnanoadvectvelocity", "HNanoAdvectVelocity", 1, 1, "OP_FLAG_GENERATOR", "AdvectIndexGridVelocity", 2, hns_parms_hnanoadvectvelocity},
    /* src/SOP/ProjectNonDivergent/SOP_VDBProjectNonDivergent.cpp:61-65 */
    {"hnanoprojectnondivergent", "HNanoProjectNonDivergent", 1, 1, "OP_FLAG_GENERATOR", "ProjectNonDivergent, Divergence", 4, hns_parms_hnanoprojectnondivergent},
    /* src/SOP/ReadWrite/SOP_VDBFromGrid.cpp:61-64 */
    {"hnanofromgrid", "HNanoFromGrid", 2, 2, "OP_FLAG_GENERATOR", "", 3, hns_parms_hnanofromgrid},
};
#define HNS_SOP_OPERATOR_COUNT 5

/* How SOP_HNanoSolverVerb::cook maps its parameters onto the C ABI (SOP_HNanoSolver.cpp:186-199,241-256):
 *   padding            -> dilation of the velocity topology by `padding` voxels, NN_FACE_EDGE_VERTEX, tiles ignored,
 *                         then union with the SDF topology: hns_dilate_leaves + hns_union_leaves (include/hns.h)
 *   timestep           -> dt;  iterations -> iterations
 *   expansion_rate     -> hns_combustion_params.expansionRate      temperature_gain -> .temperatureRelease
 *   buoyancy_strength  -> .buoyancyStrength    ambient_temp -> .ambientTemp    vorticity -> .vorticityScale
 *   factor_scale       -> .factorScale
 *   voxelSize          =  primary velocity grid's voxelSize()[0] (:184) */
#endif
