// dmpp_decision.hpp — C++ host surface, part 3: class CDecision (Decision.h:6-109).  The reference
// exposes only Instance() and startCDecisionThread(); decide() is the factored body of one pass of
// CDecisionThread for the in-scope scenes (Decision.cpp:172-205): corridor queries, the lateral-offset
// sweep, the lane-change rule tree (Decision.cpp:1011-1772), the junction handlers.
#pragma once
#include "dmpp_planning.hpp"

class CDecision : public CShare {
public:
    static CDecision& Instance();                 // Decision.cpp:36-40
    BYTE startCDecisionThread();                  // Decision.cpp:42-51: no thread; returns 1

    // junction_polyline: decision_InterMapData[...] for pos 1/2 (Decision.cpp:348); stub_attribute: Decision.cpp:385
    DecisionOut decide(const LocationOut& location, const vector<ObPoint>& obstacles,
                        const vector<GlobalPoint2D>& junction_polyline = {}, int stub_attribute = 0,
                        Path_Obs around[6] = nullptr, double period_last_ms = 100.0 /* z_period_last, Decision.cpp:137 */);
    const SceneState& State() const { return m_state; }
    void Reset();
private:
    CDecision();
    ~CDecision() {}
    SceneState m_state;             // z_behavior, light, obstacle counters ... (Decision.h:27-44, Decision.cpp:915-917)
};
